"""19x19 fused rollouts (256 plies per launch) on MID-SIZE batches (2 048 .. 16 384 games): which kernel family should serve them?
A/B build (LIB=...): the latency-shaped kernel (GG_AB_LAT_MAX), the two-board per-ply kernel, k_rollout4 with 2 .. 16 boards per wave
(GG_AB_MULTI_MIN=1 GG_AB_NB=n) - ms per launch, with state digests (every variant must leave identical states).
    LIB=ab_tmp/libgymgo_ab.so python tools/exp/mid_batch.py"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['LIB'])
from gymgo_amd import gogame
N, F = int(os.environ.get('GGN', '19')), int(os.environ.get('PLIES', '256'))
SIZES = [int(x) for x in os.environ.get('SIZES', '1024,2048,3072,4096,6144,8192,12288,16384,32768').split(',')]


def rate(B, env, reps=6):
    for k in ('GG_AB_LAT_MAX', 'GG_AB_LAT_PLIES', 'GG_AB_MULTI_MIN', 'GG_AB_NB'):
        os.environ.pop(k, None)
    os.environ.update(env)
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927, 0, 'cuda')
    ch = max(1, B // 16)
    for g in range(1, 16):
        gogame.batch_rollout(st[g * ch:(g + 1) * ch], rng[g * ch:(g + 1) * ch], g * (8 if N <= 9 else 20 if N <= 13 else 40), True)
    for _ in range(12):
        gogame.batch_rollout(st, rng, F, True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        gogame.batch_rollout(st, rng, F, True)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, hashlib.sha1(st.cpu().numpy().tobytes()).hexdigest()[:8]


variants = [('lat', {'GG_AB_LAT_MAX': '1000000', 'GG_AB_LAT_PLIES': '1'}),
            ('two-board', {'GG_AB_LAT_MAX': '0', 'GG_AB_MULTI_MIN': '100000000'})]
for nb in (2, 4, 8, 16):
    variants.append(('rollout4 nb=%d' % nb, {'GG_AB_LAT_MAX': '0', 'GG_AB_MULTI_MIN': '1', 'GG_AB_NB': str(nb)}))
print('N %d, %d plies per launch: ms per launch (steps/s)' % (N, F))
for B in SIZES:
    row, digs = [], set()
    for name, env in variants:
        ms, dg = rate(B, env)
        digs.add(dg)
        row.append('%s %.4f (%.2e)' % (name, ms, B * F / ms * 1e3))
    print('B %6d: ' % B + ' | '.join(row) + (' | same digest' if len(digs) == 1 else ' | DIGESTS DIFFER'), flush=True)
