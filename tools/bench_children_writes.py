import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gymgo_amd import gogame
N, B = 19, 8192
S = 6 * N * N
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps
st = torch.zeros((B, 6, N, N), dtype=torch.uint8, device='cuda')
st[:, 3] = 1      # every point invalid -> 361 zero slots + the pass child
out = {}
t = timed(lambda: gogame.batch_children(st))
out['all_invalid'] = {'ms': t * 1e3, 'GBps': B * 362 * S / t / 1e9}
st[:, 3] = 0      # empty board: 362 legal children, all computed
t = timed(lambda: gogame.batch_children(st))
out['all_valid_empty_board'] = {'ms': t * 1e3, 'GBps': B * 362 * S / t / 1e9}
buf = torch.empty(B * 362 * S, dtype=torch.uint8, device='cuda')
t = timed(lambda: buf.zero_())
out['torch_memset'] = {'ms': t * 1e3, 'GBps': buf.numel() / t / 1e9}
src = torch.empty_like(buf)
t = timed(lambda: buf.copy_(src))
out['torch_copy'] = {'ms': t * 1e3, 'GBps_written': buf.numel() / t / 1e9}
print(json.dumps(out, indent=1))
