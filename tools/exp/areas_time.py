"""gg_batch_areas: rate at the config sizes (the oracle comparison for every N lives in tests/test_gpu_parity.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gymgo_amd import gogame
for N, B in ((19, 65536), (19, 8192), (13, 65536), (9, 4096), (9, 65536), (5, 1000)):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 5)
    ch = max(1, B // 8)
    for g in range(8):
        gogame.batch_rollout(st[g * ch:(g + 1) * ch], rng[g * ch:(g + 1) * ch], 5 + g * (N * N) // 6, False)
    out = (torch.empty(B, dtype=torch.int32, device='cuda'), torch.empty(B, dtype=torch.int32, device='cuda'))
    gogame.batch_areas(st, out=out); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): gogame.batch_areas(st, out=out)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print('%2dx%-2d B %6d  %.1f us  %.3e boards/s  planes 0/1 at %.2f TB/s' % (N, N, B, us, B / us * 1e6, B * 2 * N * N / us / 1e6), flush=True)
