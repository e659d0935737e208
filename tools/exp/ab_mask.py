"""gg_batch_invalid_mask / gg_batch_track_states on the stationary mix (19x19; GGB boards) for ONE library (LIB=...)."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['LIB'])
from gymgo_amd import gogame
N = int(os.environ.get('GGN', 19))
for B in (65536, 131072, 49152):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 3)
    ch = B // 16
    for g in range(16):
        gogame.batch_rollout(st[g * ch:(g + 1) * ch], rng[g * ch:(g + 1) * ch], 5 + g * (2 * N), True)
    res = []
    for name, f in (('mask', lambda: gogame._invalid_mask_dev(st)), ('track', lambda: gogame.batch_track(st))):
        out = f(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(32): out = f()
        b.record(); torch.cuda.synchronize()
        res.append('%s %.2f us %s' % (name, a.elapsed_time(b) / 32 * 1e3, hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:8]))
    print('%-24s B %6d  %s' % (os.environ.get('LIB', 'shipped'), B, '   '.join(res)), flush=True)
