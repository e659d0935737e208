"""gg_batch_children_offsets alone (counts + launch order + offsets) on 8 192 mid-game 19x19 parents: us per call for one library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['LIB'])
from gymgo_amd import gogame
B, N = int(os.environ.get('GGB', 8192)), 19
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 7)
gogame.batch_rollout(st, rng, 150, True)
offs = torch.empty(B + 1, dtype=torch.int32, device='cuda'); order = torch.empty(B, dtype=torch.int32, device='cuda')
lib = _lib.lib()
def call(o):
    _lib.check(lib.gg_batch_children_offsets(st.data_ptr(), offs.data_ptr(), o, B, N, _lib.stream_ptr('cuda')), 'offsets')
for o, name in ((order.data_ptr(), 'with order'), (0, 'offsets only')):
    for _ in range(5): call(o)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): call(o)
    b.record(); torch.cuda.synchronize()
    print('%s B %d %s: %.2f us per call, total %d' % (os.environ.get('LIB', 'shipped'), B, name, a.elapsed_time(b) / 50 * 1e3, int(offs[B])))
