"""fused rollout, long launches (ramps negligible): steps/s vs batch size"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import gogame
N, F = 19, 2048
for B in (16384, 32768, 49152, 65536, 69632, 81920, 98304, 131072, 196608, 262144):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
    ch = B // 16
    for g in range(1, 16):
        gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 40, True)
    gogame.batch_rollout(st, rng, 512, True); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(2): gogame.batch_rollout(st, rng, F, True)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 2
    print('B %7d (%.2f rounds of 4 096 waves)  %.3f ms/launch  %.3e steps/s   %.3f us per ply-round' % (B, B / 65536.0, ms, B * F / ms * 1e3, ms * 1e3 / F / max(1.0, B / 65536.0)), flush=True)
