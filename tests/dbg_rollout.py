import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gymgo_amd import gogame
from oracle import c_oracle
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = 8
rng = gogame.rng_seed(B, 1)
rng_np = c_oracle.rng_seed(1, B)
print('seed eq', np.array_equal(rng.cpu().numpy().view(np.uint64), rng_np))
st = torch.zeros((B, 6, N, N), dtype=torch.uint8, device='cuda')
want = np.zeros((B, 6, N, N), np.uint8)
for ply in range(30):
    acts = gogame.batch_sample_actions(st, rng.clone())
    last = torch.full((B,), -7, dtype=torch.int32, device='cuda')
    gogame.batch_rollout(st, rng, 1, True, last, None)
    want, rng_np, wl = c_oracle.batch_rollout(want, rng_np, 1, True)
    ok_r = np.array_equal(rng.cpu().numpy().view(np.uint64), rng_np)
    print(ply, 'sample', acts.cpu().numpy(), 'rollout', last.cpu().numpy(), 'oracle', wl, 'rng_ok', ok_r,
          'state_ok', np.array_equal(st.cpu().numpy(), want))
    if not np.array_equal(st.cpu().numpy(), want):
        break
