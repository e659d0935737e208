import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', os.environ['LIB'])
from gymgo_amd import gogame
B, N = 8192, 19
for plies in (6, 20, 60, 250, 400):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 7)
    gogame.batch_rollout(st, rng, plies, True)
    kids = torch.empty((B, N * N + 1, 6, N, N), dtype=torch.uint8, device='cuda')
    gogame.batch_children(st, out=kids); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(8): gogame.batch_children(st, out=kids)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 8
    print('%-18s plies %3d  %.3f ms  %.3e parents/s  %.2f TB/s  digest %d' % (os.environ.get('LIB', 'shipped'), plies, ms, B / ms * 1e3, B * 786258 / ms / 1e9, int(kids.sum())))
    del kids
