"""k_next_states16: grid / split variants (A/B library, GG_AB_NS16*), 19x19 x 65 536 stationary mix"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CONFIGS = [{'GG_AB_NS16': '0'}, {}, {'GG_AB_NS16_GRID': '4096'}, {'GG_AB_NS16_GRID': '8192'}]
if len(sys.argv) == 1:
    for cfg in CONFIGS:
        r = subprocess.run([sys.executable, __file__, 'run'], env=dict(os.environ, **cfg), capture_output=True, text=True)
        print(cfg, r.stdout.strip() or r.stderr[-500:], flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', 'libgymgo_ab.so')
from gymgo_amd import gogame
held = {k: os.environ.pop(k) for k in list(os.environ) if k.startswith('GG_AB_')}
def ev(fn, reps=32):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
res = []
for N, B in ((19, 65536), (13, 65536), (9, 65536)):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
    ch = B // 16
    for g in range(1, 16):
        gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * (N * N // 9), True)
    gogame.batch_rollout(st, rng, 256 * 7, True)
    acts = gogame.batch_sample_actions(st, rng)
    out, status = torch.empty_like(st), torch.empty(B, dtype=torch.int32, device='cuda')
    os.environ.update(held)
    res.append('%dx%d %.1f us' % (N, N, ev(lambda: gogame.batch_next_states(st, acts, check=False, out=out, status=status))))
    for k in held: os.environ.pop(k)
print('  '.join(res))
