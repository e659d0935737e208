"""gg_batch_next_states on the stationary mix (65 536 / 131 072 / 49 152 boards of 19x19; 13x13, 9x9), shipped library"""
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
for N, B in ((19, 65536), (19, 131072), (19, 49152), (19, 262144), (13, 65536), (9, 65536)):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
    ch = B // 16
    for g in range(1, 16):
        gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * (N * N // 9), True)
    gogame.batch_rollout(st, rng, 256 * 7, True)
    acts = gogame.batch_sample_actions(st, rng)
    out, status = torch.empty_like(st), torch.empty(B, dtype=torch.int32, device='cuda')
    us = ev(lambda: gogame.batch_next_states(st, acts, check=False, out=out, status=status))
    print('%dx%d B %6d: %.1f us  %.3e steps/s  frac of %d B/step roofline %.3f' % (N, N, B, us, B / us * 1e6, 12 * N * N + 4, (12 * N * N + 4) * B / us * 1e6 / 8e12), flush=True)
