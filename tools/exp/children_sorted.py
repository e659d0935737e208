"""Does the ORDER of the parents matter to the padded gg_batch_children?  The same 8 192 stationary-mix parents as they come,
sorted by falling number of legal moves (heaviest first), by rising number, and shuffled."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import gogame
N, B = 19, 8192
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 77)
for g in range(1, 16):
    gogame.batch_rollout(st[g * 512:(g + 1) * 512], rng[g * 512:(g + 1) * 512], g * 40, True)
gogame.batch_rollout(st, rng, 300, True)
kids = torch.empty((B, N * N + 1, 6, N, N), dtype=torch.uint8, device='cuda')
cnt = (st[:, 3].reshape(B, -1) == 0).sum(1)
def t(parents):
    gogame.batch_children(parents, out=kids); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(8): gogame.batch_children(parents, out=kids)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 8
for name, idx in (('as they come', torch.arange(B, device='cuda')), ('falling count', torch.argsort(cnt, descending=True)),
                  ('rising count', torch.argsort(cnt)), ('shuffled', torch.randperm(B, device='cuda')), ('as they come', torch.arange(B, device='cuda'))):
    print('%-14s %.4f ms' % (name, t(st[idx].contiguous())), flush=True)
