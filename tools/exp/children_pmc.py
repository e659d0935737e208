"""A few launches of gg_batch_children (padded) and gg_batch_children_compact on 8 192 stationary-mix parents, for rocprofv3
--pmc passes (tools/exp/README.md)."""
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
buf = kids.view(-1, 6, N, N)
offs = torch.empty(B + 1, dtype=torch.int32, device='cuda')
for _ in range(4):
    gogame.batch_children(st, out=kids)
for _ in range(4):
    gogame.batch_children(st, padded=False, out=buf, offsets=offs)
torch.cuda.synchronize()
print('done', int(offs[B]))
