"""k_rollout5 on 9x9 / 13x13 boards (A/B build, GG_AB_R5_SMALL=1 admits the sizes, GG_AB_R5 = 0 / 1 forces the choice): ms per
launch of 256 plies against the kernels that serve the call otherwise (k_rollout_lat / k_rollout4), identical digests.
    LIB=ab_tmp/libgg_small.so GGN=9 python tools/exp/r5_small.py"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, os.environ['LIB'])
from gymgo_amd import gogame
os.environ['GG_AB_R5_SMALL'] = '1'
N = int(os.environ.get('GGN', '9'))
F = 256
for B in (32768, 40960, 49152, 65536, 98304, 131072):
    row = []
    for r5 in (0, 1, 0, 1):
        os.environ['GG_AB_R5'] = '0'
        st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927, 0, 'cuda')
        ch = B // 16
        for g in range(1, 16):
            gogame.batch_rollout(st[g * ch:(g + 1) * ch], rng[g * ch:(g + 1) * ch], g * (N * N // 8), True)
        gogame.batch_rollout(st, rng, 512, True)
        os.environ['GG_AB_R5'] = '1' if r5 else '0'
        for _ in range(4): gogame.batch_rollout(st, rng, F, True)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(12): gogame.batch_rollout(st, rng, F, True)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 12
        row.append('%s %.4f ms (%.3e) %s' % ('r5' if r5 else 'r4/lat', ms, B * F / ms * 1e3, hashlib.sha1(st.cpu().numpy().tobytes()).hexdigest()[:8]))
    print('%dx%d B %6d: ' % (N, N, B) + ' | '.join(row), flush=True)
