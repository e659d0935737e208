"""Kernel durations of one-ply launches at config 2's size under rocprofv3 --kernel-trace --stats: byte planes (k_rollout2) and
tracked boards (k_rollout_lat), 200 launches each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import gogame
N, B = int(os.environ.get('GGN', 9)), int(os.environ.get('GGB', 4096))
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
gogame.batch_rollout(st, rng, 40, True)
tr = gogame.batch_track(st)
for _ in range(200): gogame.batch_rollout(st, rng, 1, True)
for _ in range(200): gogame.batch_rollout_tracked(tr, rng, 1, True)
for _ in range(200): gogame.batch_rollout(st, rng, 2, True)
torch.cuda.synchronize()
print('done')
