"""gg_batch_sample_actions: rate at 19x19 x 65 536 and 9x9 x 4 096 (mid-game boards)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import gogame
for N, B in ((19, 65536), (13, 65536), (9, 4096), (19, 1)):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 5)
    gogame.batch_rollout(st, rng, N * N // 2, False)
    gogame.batch_sample_actions(st, rng); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): gogame.batch_sample_actions(st, rng)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print('%2dx%-2d B %6d  %.1f us  %.3e boards/s' % (N, N, B, us, B / us * 1e6), flush=True)
