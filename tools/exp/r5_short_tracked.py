"""k_rollout5 against the kernels that serve short launches on TRACKED boards (A/B build, GG_AB_R5 = 0 / 1): ms per launch of
1 .. 16 plies at 65 536 games - what a one-ply step on the thirty-two-board layout would start from.
    LIB=tools/exp/libgymgo_ab.so python tools/exp/r5_short_tracked.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['LIB'])
from gymgo_amd import gogame
N, B = 19, 65536
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927, 0, 'cuda')
ch = B // 16
os.environ['GG_AB_R5'] = '0'
for g in range(1, 16):
    gogame.batch_rollout(st[g * ch:(g + 1) * ch], rng[g * ch:(g + 1) * ch], g * 40, True)
gogame.batch_rollout(st, rng, 1024, True)
tr = gogame.batch_track(st)
for F in (1, 2, 4, 8, 16):
    row = []
    for r5 in (0, 1, 0, 1):
        os.environ['GG_AB_R5'] = str(r5)
        for _ in range(20): gogame.batch_rollout_tracked(tr, rng, F, True)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100): gogame.batch_rollout_tracked(tr, rng, F, True)
        b.record(); torch.cuda.synchronize()
        row.append('%s %.2f us' % ('r5' if r5 else 'r4', a.elapsed_time(b) * 10))
    print('tracked 65536 x %2d plies: ' % F + ' | '.join(row), flush=True)
