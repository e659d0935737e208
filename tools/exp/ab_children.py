"""gg_batch_children (config 5: 8 192 parents of 19x19, 362 slots each) for ONE library (LIB=<path relative to the repo root>)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['LIB'])
from gymgo_amd import gogame
N, B = 19, 8192
kids = torch.empty((B, N * N + 1, 6, N, N), dtype=torch.uint8, device='cuda')
lib = _lib.lib()
res = []
for plies in (20, 150, 400):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 77)
    gogame.batch_rollout(st, rng, plies, False)
    def f(): _lib.check(lib.gg_batch_children(_lib.dev_ptr(st, torch.uint8, 's'), _lib.dev_ptr(kids, torch.uint8, 'k'), B, N, 0, _lib.stream_ptr(st.device)), 'children')
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(8): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 8
    res.append('%d plies %.3f ms %.2f TB/s' % (plies, ms, B * 786258 / ms / 1e9))
import hashlib
print('%-24s %s  digest %s' % (os.environ.get('LIB', 'shipped'), '   '.join(res), hashlib.sha1(kids[:64].cpu().numpy().tobytes()).hexdigest()[:10]), flush=True)
