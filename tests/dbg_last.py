import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gymgo_amd import gogame
from oracle import c_oracle
N, B, auto = 19, 131, False
rng = gogame.rng_seed(B, 7 + N); rng_np = c_oracle.rng_seed(7 + N, B)
st = torch.zeros((B, 6, N, N), dtype=torch.uint8, device="cuda"); want = np.zeros((B, 6, N, N), np.uint8)
la = torch.empty(B, dtype=torch.int32, device="cuda")
for plies in (8, 9):
    gogame.batch_rollout(st, rng, plies, auto, la, None)
    want, rng_np, wl = c_oracle.batch_rollout(want, rng_np, plies, auto)
    bad = np.flatnonzero(la.cpu().numpy() != wl)
    print(plies, 'bad', bad[:10], la.cpu().numpy()[bad[:10]], wl[bad[:10]], 'states equal', np.array_equal(st.cpu().numpy(), want))
