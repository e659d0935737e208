import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gymgo_amd import gogame
B, N = 65536, 19
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 3)
gogame.batch_rollout(st, rng, 250, True)
acts = gogame.batch_sample_actions(st, rng)
for _ in range(6):
    out, status = gogame.batch_next_states(st, acts, check=False)
torch.cuda.synchronize()
