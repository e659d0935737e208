"""Config 2 (9x9 x 4 096 games): the per-ply-analysis rollout kernel (shipped dispatch) vs the multi-ply kernel forced
on, for every boards-per-wave value.  A/B library only (GG_AB_MULTI_MIN, GG_AB_NB)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    for size, B in ((9, 4096), (9, 8192), (13, 4096), (19, 4096), (19, 16384)):
        for nb in ('shipped', '2', '4', '6', '8'):
            env = dict(os.environ)
            if nb != 'shipped':
                env.update(GG_AB_MULTI_MIN='1', GG_AB_NB=nb)
            subprocess.run([sys.executable, __file__, str(size), str(B), nb], env=env)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', 'libgymgo_ab.so')
from gymgo_amd import gogame
N, B, tag = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
F = 256
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
ch = B // 16
for g in range(1, 16):
    gogame.batch_rollout(st[g * ch:(g + 1) * ch], rng[g * ch:(g + 1) * ch], g * 8, True)
gogame.batch_rollout(st, rng, F, True)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(8):
    gogame.batch_rollout(st, rng, F, True)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 8
import hashlib
print('%2dx%-2d B %6d nb %-8s ms/launch %.3f steps/s %.3e digest %s' % (N, N, B, tag, ms, B * F / ms * 1e3,
      hashlib.sha256(st.cpu().numpy().tobytes()).hexdigest()[:10]), flush=True)
