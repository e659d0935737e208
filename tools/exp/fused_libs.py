"""fused rollout (65 536 games x 256 plies per launch, stationary mix) for a list of A/B libraries (paths relative to the repo root):
    python tools/exp/fused_libs.py libgymgo_ab.so libgymgo_T2.so ...      (each measured twice, interleaved)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] != 'run':
    for rep in range(2):
        for lib in sys.argv[1:]:
            r = subprocess.run([sys.executable, __file__, 'run', lib], capture_output=True, text=True)
            print(lib, r.stdout.strip() or r.stderr[-600:], flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, sys.argv[2])
from gymgo_amd import gogame
N, B, F = 19, 65536, int(os.environ.get('PLIES', '256'))
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
ch = B // 16
for g in range(1, 16):
    gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 40, True)
gogame.batch_rollout(st, rng, 256 * 6, True)
for _ in range(3): gogame.batch_rollout(st, rng, F, True)
torch.cuda.synchronize()
reps = 24
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps): gogame.batch_rollout(st, rng, F, True)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
import hashlib
print('%.4f ms/launch  %.3e steps/s  digest %s' % (ms, B * F / ms * 1e3, hashlib.sha1(st.cpu().numpy().tobytes()).hexdigest()[:12]))
