"""Developer aid (not collected by pytest): print the first children mismatches of gg_batch_children vs the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gymgo_amd import gogame
from oracle import c_oracle

N, B, ply = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1234 + N
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, seed)
gogame.batch_rollout(st, rng, ply, auto_reset=False)
host = st.cpu().numpy(); live = host[:, 5, 0, 0] == 0
d = st[torch.from_numpy(live).cuda()].contiguous(); hp = host[live]
canon = len(sys.argv) > 5 and sys.argv[5] == '1'
kids = gogame.batch_children(d, canonical=canon).cpu().numpy()
want = c_oracle.batch_children(hp, canon)
bad = np.argwhere((kids != want).reshape(len(kids), N * N + 1, -1).any(-1))
print('mismatching (parent, action):', len(bad), 'of', len(kids) * (N * N + 1))
def show(s):
    ch = {(0, 0): '.', (1, 0): 'X', (0, 1): 'O', (1, 1): '?'}
    return [''.join(ch[(int(s[0, r, c]), int(s[1, r, c]))] for c in range(N)) for r in range(N)]
for b, a in bad[:4]:
    print('parent', b, 'turn', hp[b, 2, 0, 0], 'action', a, '= (r, c)', divmod(a, N))
    for p in range(6):
        diff = np.argwhere(kids[b, a, p] != want[b, a, p])
        if len(diff): print('  plane', p, 'differs at', diff.tolist()[:12], 'got', [int(kids[b, a, p, r, c]) for r, c in diff[:12]])
    par, w = show(hp[b]), show(want[b, a])
    inv = [''.join('#' if want[b, a, 3, r, c] else '.' for c in range(N)) for r in range(N)]
    got = [''.join('#' if kids[b, a, 3, r, c] else '.' for c in range(N)) for r in range(N)]
    for r in range(N): print('   ', par[r], ' ', w[r], ' ', inv[r], ' ', got[r])
