"""Fused rollout on a small batch for ONE library (LIB=..., GGN board size, GGB games, PLIES per launch): config 2 and friends.
With an A/B build (-DGG_AB) the dispatch can be forced: GG_AB_MULTI_MIN=1 (multi-ply kernel on any batch), GG_AB_NB=<boards per wave>."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['LIB'])
from gymgo_amd import gogame
N, B, F = int(os.environ.get('GGN', 9)), int(os.environ.get('GGB', 4096)), int(os.environ.get('PLIES', 256))
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
ch = max(1, B // 16)
for g in range(1, 16):
    gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 8, True)
for _ in range(3): gogame.batch_rollout(st, rng, F, True)
torch.cuda.synchronize()
reps = 16
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps): gogame.batch_rollout(st, rng, F, True)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
import hashlib
print('%s N %d B %d F %d MULTI_MIN %s NB %s: %.4f ms/launch %.3e steps/s digest %s' % (os.environ.get('LIB', 'shipped'), N, B, F, os.environ.get('GG_AB_MULTI_MIN'), os.environ.get('GG_AB_NB'),
      ms, B * F / ms * 1e3, hashlib.sha1(st.cpu().numpy().tobytes()).hexdigest()[:10]), flush=True)
