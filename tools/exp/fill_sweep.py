import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gymgo_amd import gogame
N, F = 19, int(os.environ.get('F', '256'))
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
    print('B %7d  waves %5d  ms/launch %.3f  steps/s %.3e' % (B, (B + 15) // 16, ms, B * F / ms * 1e3), flush=True)
for B in (16384, 32768, 49152, 61440, 65536, 69632, 81920, 98304, 131072, 196608, 262144):
    run(B)
