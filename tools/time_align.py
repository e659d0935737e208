import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gymgo_amd import gogame
B = 65536
for N in (15, 16, 17, 19):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 3)
    gogame.batch_rollout(st, rng, 200, True)
    res = {}
    for F in (1, 16):
        for _ in range(2): gogame.batch_rollout(st, rng, F, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 40 if F == 1 else 6
        e0.record()
        for _ in range(reps): gogame.batch_rollout(st, rng, F, True)
        e1.record(); torch.cuda.synchronize()
        res[F] = e0.elapsed_time(e1) / reps * 1e3
    io = res[1] - res[16] / 16
    print('N=%d S=%d (S%%16=%d): F=1 %.1f us/launch, fused %.1f us/ply -> I/O overhead %.1f us = %.0f GB/s effective' % (
        N, 6*N*N, (6*N*N) % 16, res[1], res[16] / 16, io, B * (4*N*N + 6*N*N) / io / 1e3))
