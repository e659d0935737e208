"""gg_batch_symmetry timings for ONE library (LIB=<path relative to the repo root>, default: the shipped one)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['LIB'])
from gymgo_amd import gogame


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


N = 19
for B in (8192, 65536):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 3)
    gogame.batch_rollout(st, rng, 200, True)
    S = 6 * N * N
    lib = _lib.lib()
    if B * 8 * S < 9e9:
        out8 = torch.empty((B, 8, 6, N, N), dtype=torch.uint8, device='cuda')
        t = timed(lambda: _lib.check(lib.gg_batch_symmetry(_lib.dev_ptr(st, torch.uint8, 's'), None, _lib.dev_ptr(out8, torch.uint8, 'o'), B, 6, N, _lib.stream_ptr(st.device)), 'sym'))
        print('%-24s B %6d all eight  %8.1f us  %.2f TB/s moved' % (os.environ.get('LIB', 'shipped'), B, t, B * 9 * S / t / 1e6), flush=True)
        del out8
    orient = torch.randint(0, 8, (B,), dtype=torch.int32, device='cuda')
    out1 = torch.empty_like(st)
    t = timed(lambda: _lib.check(lib.gg_batch_symmetry(_lib.dev_ptr(st, torch.uint8, 's'), _lib.dev_ptr(orient, torch.int32, 'or'), _lib.dev_ptr(out1, torch.uint8, 'o'), B, 6, N, _lib.stream_ptr(st.device)), 'sym'))
    print('%-24s B %6d one view   %8.1f us  %.2f TB/s moved' % (os.environ.get('LIB', 'shipped'), B, t, B * 2 * S / t / 1e6), flush=True)
# row-mask boards (tracked / packed): one view per board, all eight
for planes, name in ((5, 'tracked'), (3, 'packed')):
    B = 65536
    st = gogame.batch_init_state(B, 19, device='cuda'); rng = gogame.rng_seed(B, 3)
    gogame.batch_rollout(st, rng, 150, True)
    rows = gogame.batch_track(st) if planes == 5 else gogame.batch_pack(st)
    orient = torch.randint(0, 8, (B,), dtype=torch.int32, device='cuda')
    for label, o, nb in (('one view', orient, B), ('no rotation', orient & 3, B), ('all eight', None, 8192)):
        r = rows[:nb]; oo = None if o is None else o[:nb]
        out = gogame.batch_symmetry_rows(r, 19, oo); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(16): out = gogame.batch_symmetry_rows(r, 19, oo)
        e1.record(); torch.cuda.synchronize()
        import hashlib
        print('%-24s rows %-8s B %6d %-12s %8.1f us  digest %s' % (os.environ.get('LIB', 'shipped'), name, nb, label, e0.elapsed_time(e1) / 16 * 1e3,
              hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:10]), flush=True)
