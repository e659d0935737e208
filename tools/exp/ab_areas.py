"""gg_batch_areas on the stationary mix of 65 536 boards (19x19; GGN for another size) for ONE library (LIB=...)."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['LIB'])
from gymgo_amd import gogame
N, B = int(os.environ.get('GGN', 19)), int(os.environ.get('GGB', 65536))
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 3)
ch = B // 16
for g in range(16):
    gogame.batch_rollout(st[g * ch:(g + 1) * ch], rng[g * ch:(g + 1) * ch], 5 + g * (2 * N), True)
out = (torch.empty(B, dtype=torch.int32, device='cuda'), torch.empty(B, dtype=torch.int32, device='cuda'))
gogame.batch_areas(st, out=out); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(64): gogame.batch_areas(st, out=out)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 64 * 1e3
print('%-24s N %d B %d: %.2f us %.3e boards/s digest %s' % (os.environ.get('LIB', 'shipped'), N, B, us, B / us * 1e6,
      hashlib.sha1(torch.stack(out).cpu().numpy().tobytes()).hexdigest()[:10]), flush=True)
