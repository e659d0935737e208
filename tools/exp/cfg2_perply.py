"""Config 2 (4 096 games of 9x9), ONE ply per launch on byte planes: gg_batch_rollout through the API and as a hipGraph of 64
launches, for ONE library (LIB=<path relative to the repo root>, default: the shipped one); GGN / GGB as ab_small.py."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['LIB'])
from gymgo_amd import gogame
N, B = int(os.environ.get('GGN', 9)), int(os.environ.get('GGB', 4096))
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
ch = max(1, B // 16)
for g in range(1, 16):
    gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 8, True)
gogame.batch_rollout(st, rng, 64, True)
def rate(fn, units, reps):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    return units / ms * 1e3, ms
out = []
for F in (1, 2):
    r, ms = rate(lambda: gogame.batch_rollout(st, rng, F, True), B * F, 64)
    side = torch.cuda.Stream(); graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side): gogame.batch_rollout(st, rng, F, True)
    side.synchronize()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(64): gogame.batch_rollout(st, rng, F, True)
    rg, msg = rate(graph.replay, B * F * 64, 16)
    out.append('F %d: API %.2f us %.3e steps/s | graph %.2f us per launch %.3e steps/s' % (F, ms * 1e3, r, msg * 1e3 / 64, rg))
print('%s N %d B %d: %s digest %s' % (os.environ.get('LIB', 'shipped'), N, B, ' ; '.join(out), hashlib.sha1(st.cpu().numpy().tobytes()).hexdigest()[:10]), flush=True)
