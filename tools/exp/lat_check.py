"""The latency-shaped multi-ply kernel (gg_lat.h) on the GPU box: parity against the C oracle over board sizes, batch sizes
and launch lengths that gg_batch_rollout sends to it, then its rate against the kernels it replaces.
  LIB=<path>   library under test (default: the shipped one)
  MODE=parity|time|sweep (default: all)
With an A/B build GG_AB_LAT_MAX=<games per CU> moves the take-over point (0: never).
"""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gymgo_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['LIB'])
from gymgo_amd import gogame
from oracle import c_oracle

dev = 'cuda'
MODE = os.environ.get('MODE', 'all')


def parity():
    bad = 0
    total_plies = 0
    for N, B, launches in ((9, 4096, (2, 30, 64, 104, 256)), (9, 1, (3, 50)), (9, 3, (64, 64)), (9, 5, (64, 200)), (9, 1001, (7, 120, 256)),
                           (13, 1023, (2, 100, 300)), (13, 6, (200,)), (19, 511, (2, 150, 400)), (19, 2, (300, 300)), (19, 3, (600,)),
                           (5, 77, (40, 100)), (7, 130, (64, 64, 64)), (2, 9, (20,)), (3, 10, (30,)), (11, 100, (250,)), (16, 33, (300,)),
                           (8, 64, (128,)), (12, 64, (200,)), (14, 40, (300,)), (19, 4095, (64,)), (13, 4000, (64,)), (9, 8191, (32,))):
        for auto_reset in (True, False):
            st = gogame.batch_init_state(B, N, device=dev)
            rng = gogame.rng_seed(B, 4242 + N, 0, dev)
            ref, ref_rng = st.cpu().numpy(), rng.cpu().numpy().view(np.uint64)
            steps = torch.zeros(B, dtype=torch.int64, device=dev)
            for F in launches:
                last = torch.full((B,), -7, dtype=torch.int32, device=dev)
                lib = _lib.lib()
                _lib.check(lib.gg_batch_rollout(_lib.dev_ptr(st, torch.uint8, 's'), rng.data_ptr(), last.data_ptr(), steps.data_ptr(),
                                                B, N, F, 1 if auto_reset else 0, _lib.stream_ptr(st.device)), 'rollout')
                torch.cuda.synchronize()
                ref, ref_rng, ref_last = c_oracle.batch_rollout_mt(ref, ref_rng, F, auto_reset)
                got, grng, glast = st.cpu().numpy(), rng.cpu().numpy().view(np.uint64), last.cpu().numpy()
                total_plies += B * F
                if not (np.array_equal(got, ref) and np.array_equal(grng, ref_rng) and np.array_equal(glast, ref_last)):
                    bad += 1
                    wrong = np.flatnonzero((got != ref).reshape(B, -1).any(axis=1))
                    print('MISMATCH N %d B %d F %d auto_reset %s: %d boards differ (first %s), rng %d, last %d' %
                          (N, B, F, auto_reset, len(wrong), wrong[:6].tolist(), int((grng != ref_rng).sum()), int((glast != ref_last).sum())), flush=True)
                    if len(wrong):
                        b = int(wrong[0])
                        d = np.argwhere(got[b] != ref[b])
                        print('  board %d: planes %s cells %s' % (b, sorted(set(d[:, 0].tolist())), d[:6].tolist()), flush=True)
                    st.copy_(torch.from_numpy(ref)); rng.copy_(torch.from_numpy(ref_rng.view(np.int64)).view(rng.dtype))
    print('parity: %d mismatching launches, %.2e oracle-checked plies' % (bad, total_plies), flush=True)
    return bad


def rate(N, B, F, reps=8):
    st = gogame.batch_init_state(B, N, device=dev); rng = gogame.rng_seed(B, 20260927, 0, dev)
    ch = max(1, B // 16)
    for g in range(1, 16):
        gogame.batch_rollout(st[g * ch:(g + 1) * ch], rng[g * ch:(g + 1) * ch], g * (8 if N <= 9 else 20 if N <= 13 else 40), True)
    for _ in range(3):
        gogame.batch_rollout(st, rng, F, True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        gogame.batch_rollout(st, rng, F, True)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    return ms, B * F / ms * 1e3, hashlib.sha1(st.cpu().numpy().tobytes()).hexdigest()[:10]


def timing():
    for N, B, F in ((9, 4096, 256), (9, 4096, 64), (9, 4096, 8), (9, 4096, 2)):
        ms, r, dg = rate(N, B, F)
        print('%s LAT_MAX %s: N %d B %d F %d: %.4f ms/launch %.3e steps/s digest %s' % (os.environ.get('LIB', 'shipped'), os.environ.get('GG_AB_LAT_MAX'), N, B, F, ms, r, dg), flush=True)


def sweep():
    """A/B build only (GG_AB_LAT_MAX / GG_AB_LAT_PLIES are read at every call): the new kernel forced on / off per shape."""
    def both(N, B, F, reps):
        out = []
        for lat in ('1000000', '0'):
            os.environ['GG_AB_LAT_MAX'] = lat
            os.environ['GG_AB_LAT_PLIES'] = '1'
            ms, r, dg = rate(N, B, F, reps=reps)
            out.append((ms, r, dg))
        print('N %2d B %6d F %3d: lat %.4f ms %.3e steps/s | old %.4f ms %.3e steps/s | x%.2f %s' %
              (N, B, F, out[0][0], out[0][1], out[1][0], out[1][1], out[0][1] / out[1][1], 'same digest' if out[0][2] == out[1][2] else 'DIGESTS DIFFER'), flush=True)
    for N, sizes in ((9, (256, 1024, 2048, 4096, 8192, 16384, 32768, 65536)), (13, (256, 1024, 2048, 4096, 8192, 16384, 32768)), (19, (256, 1024, 2048, 4096, 8192, 16384, 32768)),
                     (7, (4096, 16384)), (5, (4096,))):
        for B in sizes:
            both(N, B, 256, 4)
    for N, B in ((9, 4096), (9, 1024), (13, 4096), (19, 4096), (19, 1024)):
        for F in (1, 2, 4, 16):
            both(N, B, F, 32)


def rate_graph(N, B, F, nodes=32, reps=8):
    """us per launch inside a hipGraph of `nodes` launches (no host launch cost)."""
    st = gogame.batch_init_state(B, N, device=dev); rng = gogame.rng_seed(B, 20260927, 0, dev)
    ch = max(1, B // 16)
    for g in range(1, 16):
        gogame.batch_rollout(st[g * ch:(g + 1) * ch], rng[g * ch:(g + 1) * ch], g * (8 if N <= 9 else 20 if N <= 13 else 40), True)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        gogame.batch_rollout(st, rng, F, True)
    side.synchronize()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(nodes):
            gogame.batch_rollout(st, rng, F, True)
    graph.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        graph.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * nodes) * 1e3, hashlib.sha1(st.cpu().numpy().tobytes()).hexdigest()[:10]


def fsweep():
    """A/B build only: byte-plane launches of 1 .. 16 plies as hipGraph nodes, the latency-shaped kernel forced on / off."""
    for N, B in ((9, 1024), (9, 4096), (9, 8192), (13, 1024), (13, 4096), (19, 512), (19, 2048), (19, 4096)):
        for F in (1, 2, 3, 4, 6, 8, 16, 64):
            out = []
            for lat in ('1000000', '0'):
                os.environ['GG_AB_LAT_MAX'] = lat
                os.environ['GG_AB_LAT_PLIES'] = '1'
                out.append(rate_graph(N, B, F))
            print('N %2d B %6d F %3d: lat %.2f us | two-board %.2f us per launch | x%.2f %s' %
                  (N, B, F, out[0][0], out[1][0], out[1][0] / out[0][0], 'same digest' if out[0][1] == out[1][1] else 'DIGESTS DIFFER'), flush=True)


def tracked_parity():
    """Tracked boards through gg_batch_rollout_tracked (lat below its take-over point, k_rollout4 above): untracked == oracle,
    and the tracked words == track(states) bit for bit (the classes are canonical)."""
    bad = 0
    for N, B, launches in ((9, 4096, (1, 2, 30, 64, 256)), (9, 5, (1, 64, 200)), (9, 1001, (1, 7, 120)), (13, 1023, (1, 2, 100, 300)),
                           (19, 511, (1, 2, 150, 400)), (19, 3, (1, 600)), (5, 77, (1, 40, 100)), (7, 130, (1, 64, 64)), (2, 9, (1, 20)),
                           (11, 100, (250,)), (16, 33, (300,)), (9, 20000, (3, 40)), (13, 9000, (3, 40)), (19, 2500, (70,))):
        for auto_reset in (True, False):
            st = gogame.batch_init_state(B, N, device=dev)
            tr = gogame.batch_track(st)
            rng = gogame.rng_seed(B, 4242 + N, 0, dev)
            ref, ref_rng = st.cpu().numpy(), rng.cpu().numpy().view(np.uint64)
            for F in launches:
                last = torch.full((B,), -7, dtype=torch.int32, device=dev)
                gogame.batch_rollout_tracked(tr, rng, F, auto_reset, last)
                ref, ref_rng, ref_last = c_oracle.batch_rollout_mt(ref, ref_rng, F, auto_reset)
                got = gogame.batch_untrack(tr).cpu().numpy()
                ok = np.array_equal(got, ref) and np.array_equal(rng.cpu().numpy().view(np.uint64), ref_rng) and np.array_equal(last.cpu().numpy(), ref_last)
                ok = ok and torch.equal(gogame.batch_track(torch.from_numpy(ref).to(dev)), tr)
                if not ok:
                    bad += 1
                    print('TRACKED MISMATCH N %d B %d F %d auto_reset %s' % (N, B, F, auto_reset), flush=True)
    print('tracked parity: %d mismatching launches' % bad, flush=True)
    return bad


def rate_tracked(N, B, F, reps=8):
    st = gogame.batch_init_state(B, N, device=dev); rng = gogame.rng_seed(B, 20260927, 0, dev)
    ch = max(1, B // 16)
    for g in range(1, 16):
        gogame.batch_rollout(st[g * ch:(g + 1) * ch], rng[g * ch:(g + 1) * ch], g * (8 if N <= 9 else 20 if N <= 13 else 40), True)
    tr = gogame.batch_track(st)
    for _ in range(3):
        gogame.batch_rollout_tracked(tr, rng, F, True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        gogame.batch_rollout_tracked(tr, rng, F, True)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    return ms, B * F / ms * 1e3, hashlib.sha1(tr.cpu().numpy().tobytes()).hexdigest()[:10]


def tsweep():
    def both(N, B, F, reps):
        out = []
        for lat in ('1000000', '0'):
            os.environ['GG_AB_LATT_MAX'] = lat
            os.environ['GG_AB_LATT_PLIES'] = '1'
            out.append(rate_tracked(N, B, F, reps=reps))
        print('tracked N %2d B %6d F %3d: lat %.4f ms %.3e steps/s | k_rollout4 %.4f ms %.3e steps/s | x%.2f %s' %
              (N, B, F, out[0][0], out[0][1], out[1][0], out[1][1], out[0][1] / out[1][1], 'same digest' if out[0][2] == out[1][2] else 'DIGESTS DIFFER'), flush=True)
    if os.environ.get('BIG'):
        for N, sizes in ((9, (65536, 262144)), (13, (32768, 65536, 131072)), (19, (16384, 32768, 65536))):
            for B in sizes:
                for F in (1, 2, 4):
                    both(N, B, F, 24)
        return
    for N, sizes in ((9, (1024, 4096, 8192, 16384, 32768)), (13, (1024, 4096, 8192, 16384)), (19, (1024, 2048, 4096, 8192))):
        for B in sizes:
            for F in (1, 4, 64, 256):
                both(N, B, F, 4 if F >= 64 else 24)


def esweep():
    """GoVecEnv.step's launch (gg_batch_env_step_tracked: drawn moves, reward `real`, with and without the observation), the
    latency-shaped kernel forced on / off (A/B build: GG_AB_LATE_MAX)."""
    big = os.environ.get('BIG')
    for N, sizes in (((9, (131072, 262144)), (13, (65536, 131072)), (19, (32768, 65536, 131072))) if big else
                     ((9, (1024, 4096, 16384, 32768, 65536)), (13, (1024, 4096, 16384, 32768)), (19, (1024, 4096, 8192, 16384)))):
        for B in sizes:
            st = gogame.batch_init_state(B, N, device=dev); rng = gogame.rng_seed(B, 20260927, 0, dev)
            ch = max(1, B // 16)
            for g in range(1, 16):
                gogame.batch_rollout(st[g * ch:(g + 1) * ch], rng[g * ch:(g + 1) * ch], g * (8 if N <= 9 else 20 if N <= 13 else 40), True)
            base = gogame.batch_track(st)
            obs = torch.empty_like(st)
            env_out = (torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B, dtype=torch.uint8, device=dev),
                       torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev))
            for with_obs in (True, False):
                res = []
                for lat in ('1000000', '0'):
                    os.environ['GG_AB_LATE_MAX'] = lat
                    tr, rg = base.clone(), rng.clone()
                    fn = lambda: gogame.batch_env_step_tracked(tr, None, rg, 7.5, 'real', True, out=env_out, states_out=obs if with_obs else None)
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(32):
                        fn()
                    b.record(); torch.cuda.synchronize()
                    res.append((a.elapsed_time(b) / 32, hashlib.sha1(tr.cpu().numpy().tobytes() + obs.cpu().numpy().tobytes()).hexdigest()[:10]))
                print('env step N %2d B %6d obs %d: lat %.2f us | k_rollout4 %.2f us | x%.2f %s' %
                      (N, B, with_obs, res[0][0] * 1e3, res[1][0] * 1e3, res[1][0] / res[0][0], 'same digest' if res[0][1] == res[1][1] else 'DIGESTS DIFFER'), flush=True)


if __name__ == '__main__':
    rc = 0
    if MODE in ('all', 'parity'):
        rc = parity()
    if MODE in ('all', 'time'):
        timing()
    if MODE in ('sweep',):
        sweep()
    if MODE in ('fsweep',):
        fsweep()
    if MODE in ('all', 'tparity'):
        rc = rc or tracked_parity()
    if MODE in ('tsweep',):
        tsweep()
    if MODE in ('esweep',):
        esweep()
    sys.exit(1 if rc else 0)
