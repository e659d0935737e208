"""gg_batch_env_step on byte planes (in place; drawn moves, both reward methods; given moves) on the stationary mix, shipped library"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import gogame
def ev(fn, reps=32):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for N, B in ((19, 65536), (19, 131072), (13, 65536), (9, 65536)):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
    ch = B // 16
    for g in range(1, 16):
        gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * (N * N // 9), True)
    gogame.batch_rollout(st, rng, 256 * 7, True)
    out = (torch.empty(B, dtype=torch.float32, device='cuda'), torch.empty(B, dtype=torch.uint8, device='cuda'),
           torch.empty(B, dtype=torch.int32, device='cuda'), torch.empty(B, dtype=torch.int32, device='cuda'))
    w = st.clone()
    r = ev(lambda: gogame.batch_env_step(w, None, rng, 7.5, 'real', True, out=out))
    hh = ev(lambda: gogame.batch_env_step(w, None, rng, 7.5, 'heuristic', True, out=out))
    print('%dx%d B %6d: env step real %.1f us (%.3e steps/s)  heuristic %.1f us' % (N, N, B, r, B / r * 1e6, hh), flush=True)
