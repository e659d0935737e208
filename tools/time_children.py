import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gymgo_amd import gogame
B, N = 8192, 19
plies = int(sys.argv[1]) if len(sys.argv) > 1 else 250
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 3)
gogame.batch_rollout(st, rng, plies, True)
out = torch.empty((B, N * N + 1, 6, N, N), dtype=torch.uint8, device='cuda')
from gymgo_amd import _lib
def run():
    code = _lib.lib().gg_batch_children(_lib.dev_ptr(st, torch.uint8, 's'), _lib.dev_ptr(out, torch.uint8, 'o'), B, N, 0, _lib.stream_ptr(st.device))
    assert code == 0
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
nvalid = float((st[:, 3].reshape(B, -1) == 0).sum(1).float().mean())
print('children plies %d: %.3f ms per %d parents, %.3e parents/s, %.2f TB/s, frac %.3f, mean legal %.1f, FULL=%s' % (
    plies, ms, B, B / ms * 1e3, B * 786258 / ms / 1e9, B * 786258 / ms / 1e9 / 8000, nvalid, os.environ.get('GG_CHILDREN_FULL', '0')))
