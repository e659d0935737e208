import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['LIB'])
from gymgo_amd import gogame
import numpy as np
N, F = 19, 256
def run(B, reps=6):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
    ch = B // 16
    for g in range(1, 16):
        gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 40, True)
    gogame.batch_rollout(st, rng, F, True); gogame.batch_rollout(st, rng, F, True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): gogame.batch_rollout(st, rng, F, True)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    import hashlib
    h = hashlib.sha256(st.cpu().numpy().tobytes()).hexdigest()[:12]
    print('%-22s B %7d ms/launch %.3f steps/s %.3e digest %s' % (os.environ.get('LIB', 'shipped'), B, ms, B * F / ms * 1e3, h), flush=True)
run(65536); run(131072, 3)
