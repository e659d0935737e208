"""fused rollout vs plies per launch at 65 536 / 131 072 games (tail + launch overhead of one launch)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import gogame
N = 19
for B in (65536, 131072):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
    ch = B // 16
    for g in range(1, 16):
        gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 40, True)
    gogame.batch_rollout(st, rng, 512, True)
    for F in (64, 256, 1024, 4096):
        reps = max(2, 4096 // F)
        gogame.batch_rollout(st, rng, F, True); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): gogame.batch_rollout(st, rng, F, True)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        print('B %7d F %5d  %.3f ms/launch  %.3e steps/s' % (B, F, ms, B * F / ms * 1e3), flush=True)
