"""k_rollout5 (gg_v5.h: 32 boards per wave, flood jobs) against the kernels that serve the same call otherwise - A/B build,
GG_AB_R5 = 0 / 1 forces the choice: every output (states, generators, last actions, played steps) must be identical, byte planes
and tracked boards, ragged batches, short and long launches, with and without auto-reset; then ms per launch at the headline's shape.
    LIB=tools/exp/libgymgo_ab.so python tools/exp/r5_check.py [check|time|all]"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['LIB'])
from gymgo_amd import gogame
N = int(os.environ.get('GGN', '19'))


def dig(*ts):
    h = hashlib.sha1()
    for t in ts:
        h.update(t.cpu().numpy().tobytes())
    return h.hexdigest()[:10]


def play(B, r5, tracked, schedule, auto_reset, seed):
    os.environ['GG_AB_R5'] = '1' if r5 else '0'
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, seed, 0, 'cuda')
    last = torch.full((B,), -7, dtype=torch.int32, device='cuda')
    done = torch.zeros(B, dtype=torch.int64, device='cuda')
    out = []
    if tracked:
        tr = gogame.batch_track(st)
        for plies in schedule:
            gogame.batch_rollout_tracked(tr, rng, plies, auto_reset, last_actions=last, steps_done=done)
            out.append(dig(tr, rng, last, done))
    else:
        for plies in schedule:
            gogame.batch_rollout(st, rng, plies, auto_reset, last_actions=last, steps_done=done)
            out.append(dig(st, rng, last, done))
    return out, int(done.sum().item())


def check():
    bad = 0
    for B in (1, 2, 31, 32, 33, 64, 1000, 4097, 65536):
        for tracked in (False, True):
            for auto_reset in (True, False):
                schedule = (1, 2, 7, 40, 300, 700, 256) if B <= 4097 else (3, 300, 900, 256)
                a, na = play(B, False, tracked, schedule, auto_reset, 1234 + B)
                b, nb_ = play(B, True, tracked, schedule, auto_reset, 1234 + B)
                ok = a == b and na == nb_
                bad += not ok
                print('B %6d %s auto_reset %d: %s  (%d steps)  %s' % (B, 'tracked' if tracked else 'bytes  ', auto_reset,
                                                                       'same' if ok else 'DIFFER', na, '' if ok else (a, b)), flush=True)
    print('CHECK', 'ok' if not bad else 'FAILED (%d)' % bad)
    return bad


def rate(B, r5, F=256, reps=12, tracked=False):
    os.environ['GG_AB_R5'] = '1' if r5 else '0'
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927, 0, 'cuda')
    ch = max(1, B // 16)
    for g in range(1, 16):
        gogame.batch_rollout(st[g * ch:(g + 1) * ch], rng[g * ch:(g + 1) * ch], g * 40, True)
    gogame.batch_rollout(st, rng, 256 * 4, True)
    if tracked:
        st = gogame.batch_track(st)
        fn = lambda: gogame.batch_rollout_tracked(st, rng, F, True)
    else:
        fn = lambda: gogame.batch_rollout(st, rng, F, True)
    for _ in range(24):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, dig(st)


def time_():
    for B in (32768, 49152, 65536, 98304, 131072):
        for F in (256,) if B != 65536 else (8, 32, 64, 256, 1024):
            row = []
            for rep in range(2):
                for r5 in (False, True):
                    ms, dg = rate(B, r5, F, reps=12 if F >= 64 else 60)
                    row.append('%s %.4f ms (%.3e) %s' % ('r5' if r5 else 'r4', ms, B * F / ms * 1e3, dg))
            print('B %6d x %4d plies: ' % (B, F) + ' | '.join(row), flush=True)
    for r5 in (False, True):
        ms, dg = rate(65536, r5, 256, tracked=True)
        print('tracked 65536 x 256: %s %.4f ms (%.3e)' % ('r5' if r5 else 'r4', ms, 65536 * 256 / ms * 1e3), flush=True)


mode = sys.argv[1] if len(sys.argv) > 1 else 'all'
rc = 0
if mode in ('check', 'all'):
    rc = check()
if mode in ('time', 'all'):
    time_()
sys.exit(1 if rc else 0)
