"""Why does bench.py's config-5 number differ from tools/bench_ops.py's on the same box?  Times gg_batch_children (8 192 parents
of 19x19) cold, right after half a second of fused rollouts, after a pause, into a preallocated and into a fresh buffer."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib, gogame
N, B = 19, 8192
lib = _lib.lib()
big = gogame.batch_init_state(65536, N, device='cuda'); brng = gogame.rng_seed(65536, 1)
gogame.batch_rollout(big, brng, 300, True)
st = big[:B]
kids = torch.empty((B, N * N + 1, 6, N, N), dtype=torch.uint8, device='cuda')


def children(out):
    _lib.check(lib.gg_batch_children(_lib.dev_ptr(st, torch.uint8, 's'), _lib.dev_ptr(out, torch.uint8, 'k'), B, N, 0, _lib.stream_ptr(st.device)), 'children')


def timed(fn, reps=8, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def show(tag, ms):
    print('%-46s %.3f ms  %.2f TB/s  %.3g parents/s  sclk %s' % (tag, ms, B * 786258 / ms / 1e9, B / ms * 1e3, torch.cuda.clock_rate()), flush=True)


show('cold, preallocated', timed(lambda: children(kids)))
show('again', timed(lambda: children(kids)))
for _ in range(250): gogame.batch_rollout(big, brng, 256, True)      # ~0.5 s of the headline kernel
show('right after 0.5 s of fused rollouts', timed(lambda: children(kids)))
show('again (12 more calls in)', timed(lambda: children(kids), warm=12))
torch.cuda.synchronize(); time.sleep(0.5)
show('after 0.5 s idle', timed(lambda: children(kids)))
show('fresh buffer per call (as gogame.batch_children)', timed(lambda: gogame.batch_children(st)))
own = st.clone()
st = own
show('parents in their own tensor', timed(lambda: children(kids)))
