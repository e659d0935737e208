"""gg_batch_invalid_mask / track on the stationary mix, shipped library"""
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
for N, B in ((19, 65536), (19, 49152), (19, 131072), (13, 65536), (9, 65536)):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
    ch = B // 16
    for g in range(1, 16):
        gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * (N * N // 9), True)
    gogame.batch_rollout(st, rng, 256 * 7, True)
    us = ev(lambda: gogame._invalid_mask_dev(st, None))
    print('%dx%d B %6d: invalid mask %.1f us  %.3e boards/s' % (N, N, B, us, B / us * 1e6), flush=True)
