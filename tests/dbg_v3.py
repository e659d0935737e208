"""Developer aid: first diverging ply of the v3 fused rollout vs the oracle, from the snake layouts or from play."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from gymgo_amd import gogame, state_utils
from oracle import c_oracle
n = int(sys.argv[1]); mode = sys.argv[2] if len(sys.argv) > 2 else 'play'
if mode == 'play':
    B = 96
    st0 = np.zeros((B, 6, n, n), np.uint8)
    r = c_oracle.rng_seed(3, B)
    st0, r, _ = c_oracle.batch_rollout(st0, r, int(sys.argv[3]) if len(sys.argv) > 3 else 30, True)
else:
    import test_gpu_adversarial as T
    st0 = T.boards(n)
    d = torch.from_numpy(st0).cuda()
    # consistent layouts as in the test: rely on oracle mask for plane 3 after removing dead groups is done in the test; here just take the mask
    st0[:, 3] = state_utils.batch_compute_invalid_moves(d, None, None).cpu().numpy()
B = len(st0)
def show(s):
    ch = {(0, 0): '.', (1, 0): 'X', (0, 1): 'O', (1, 1): '?'}
    return [''.join(ch[(int(s[0, r, c]), int(s[1, r, c]))] for c in range(n)) for r in range(n)]
for plies in (1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 16, 24, 40):
    rng = gogame.rng_seed(B, 5); rng_np = c_oracle.rng_seed(5, B)
    roll = torch.from_numpy(st0).cuda().clone()
    gogame.batch_rollout(roll, rng, plies, True)
    want, _, _ = c_oracle.batch_rollout(st0.copy(), rng_np, plies, True)
    got = roll.cpu().numpy()
    bad = np.flatnonzero((got != want).reshape(B, -1).any(1))
    print('plies', plies, 'bad boards', len(bad), bad[:10])
    if len(bad):
        i = bad[0]
        prev, _, _ = c_oracle.batch_rollout(st0.copy(), c_oracle.rng_seed(5, B), plies - 1, True)
        for p in range(6):
            dd = np.argwhere(got[i, p] != want[i, p])
            if len(dd): print('  plane', p, 'differs at', dd.tolist()[:10])
        print('  turn before', prev[i, 2, 0, 0], 'passed', prev[i, 4, 0, 0])
        a, b, c = show(prev[i]), show(want[i]), show(got[i])
        for r in range(n): print('   ', a[r], ' ', b[r], ' ', c[r], ' ', ''.join('#' if want[i, 3, r, cc] else '.' for cc in range(n)), ' ', ''.join('#' if got[i, 3, r, cc] else '.' for cc in range(n)))
        break
