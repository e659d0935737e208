"""gg_batch_sample_weighted (byte planes) and _rows (tracked boards) on 65 536 boards of 19x19 for ONE library (LIB=...)."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['LIB'])
from gymgo_amd import gogame
B, N = 65536, 19
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 3)
gogame.batch_rollout(st, rng, 150, True)
tr = gogame.batch_track(st)
res = []
for dt in (torch.float32, torch.bfloat16):
    w = (torch.rand((B, N * N + 1), device='cuda') ** 3).to(dt)
    for name, f in (('planes', lambda: gogame.batch_sample_weighted(st, w, rng)), ('rows', lambda: gogame.batch_sample_weighted_rows(tr, N, w, rng))):
        a = f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(32): a = f()
        e1.record(); torch.cuda.synchronize()
        res.append('%s %s %.2f us' % (name, str(dt).split('.')[-1], e0.elapsed_time(e1) / 32 * 1e3))
print('%-24s %s  digest %s' % (os.environ.get('LIB', 'shipped'), '  '.join(res), hashlib.sha1(a.cpu().numpy().tobytes()).hexdigest()[:10]), flush=True)
