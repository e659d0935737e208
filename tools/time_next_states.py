import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gymgo_amd import gogame
B, N = 65536, 19
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 3)
gogame.batch_rollout(st, rng, 250, True)
acts = gogame.batch_sample_actions(st, rng)
for _ in range(3): gogame.batch_next_states(st, acts, check=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): gogame.batch_next_states(st, acts, check=False)
e1.record(); torch.cuda.synchronize()
print('GG_DBG', os.environ.get('GG_DBG', '0'), 'us per launch', e0.elapsed_time(e1) / 30 * 1e3, 'steps/s %.3e' % (B / (e0.elapsed_time(e1) / 30 * 1e-3)))
