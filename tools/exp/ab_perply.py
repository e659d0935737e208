"""per-ply API rates on the stationary mix: gg_batch_next_states (+ other per-ply entry points) for one library.
   LIB=<file under tools/exp> GG_AB_V5_MIN=<boards> python tools/exp/ab_perply.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', os.environ['LIB'])
from gymgo_amd import gogame
N = 19
def rate(fn, units, reps=32):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    return units / ms * 1e3, ms * 1e3
for B in (65536, 131072, 32768):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
    ch = B // 16
    for g in range(1, 16):
        gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 40, True)
    gogame.batch_rollout(st, rng, 256 * 7, True)
    acts = gogame.batch_sample_actions(st, rng)
    nxt, status = torch.empty_like(st), torch.empty(B, dtype=torch.int32, device='cuda')
    r, us = rate(lambda: gogame.batch_next_states(st, acts, check=False, out=nxt, status=status), B)
    mid = gogame.batch_init_state(B, N, device='cuda'); rng2 = gogame.rng_seed(B, 5)
    gogame.batch_rollout(mid, rng2, 120, True)
    acts2 = gogame.batch_sample_actions(mid, rng2)
    r2, us2 = rate(lambda: gogame.batch_next_states(mid, acts2, check=False, out=nxt, status=status), B)
    print('%-20s V5_MIN %-8s B %7d next_states stationary %.1f us (%.3e/s, frac %.3f)  mid-game %.1f us (%.3e/s)' % (
        os.environ.get('LIB', 'shipped'), os.environ.get('GG_AB_V5_MIN', '-'), B, us, r, 4336 * r / 8e12, us2, r2), flush=True)
