import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gymgo_amd import gogame
from oracle import c_oracle
np.set_printoptions(linewidth=250)

def show(N, got, want, tag):
    bad = np.flatnonzero((got != want).reshape(len(got), -1).any(1))
    print(tag, 'mismatching boards', len(bad), 'of', len(got), 'first', bad[:10])
    if len(bad) == 0:
        return False
    b = bad[0]
    for p in range(6):
        if (got[b, p] != want[b, p]).any():
            print(' board', b, 'plane', p, 'diff at', np.argwhere(got[b, p] != want[b, p])[:12].tolist())
    if os.environ.get('DBG_FULL'):
        print(' got stones (1 black 2 white)\n', got[b, 0] + 2 * got[b, 1])
        print(' want stones\n', want[b, 0] + 2 * want[b, 1])
        print(' got inv\n', got[b, 3]); print(' want inv\n', want[b, 3])
    print(' flags got', got[b, 2, 0, 0], got[b, 4, 0, 0], got[b, 5, 0, 0], 'want', want[b, 2, 0, 0], want[b, 4, 0, 0], want[b, 5, 0, 0])
    return True

def tracked_run(N, B, plies_list, seed=5):
    st = torch.zeros((B, 6, N, N), dtype=torch.uint8, device='cuda')
    rng = gogame.rng_seed(B, seed); want = np.zeros((B, 6, N, N), np.uint8); wr = c_oracle.rng_seed(seed, B)
    tr = gogame.batch_track(st)
    tot = 0
    for p in plies_list:
        gogame.batch_rollout_tracked(tr, rng, p, True)
        want, wr, _ = c_oracle.batch_rollout(want, wr, p, True)
        tot += p
        got = gogame.batch_untrack(tr).cpu().numpy()
        if show(N, got, want, 'tracked N=%d B=%d after %d plies' % (N, B, tot)):
            return False
        t2 = gogame.batch_track(gogame.batch_untrack(tr))
        if not torch.equal(t2, tr):
            d = (t2 != tr).nonzero()
            print(' class rows differ', d[:10].tolist(), 'W', tr.shape[1]); return False
    return True

def byte_run(N, B, plies_list, seed=7):
    st = torch.zeros((B, 6, N, N), dtype=torch.uint8, device='cuda')
    rng = gogame.rng_seed(B, seed)
    idx = np.arange(0, B, 16); want = np.zeros((len(idx), 6, N, N), np.uint8)
    wr = np.array([c_oracle.lib().gg_oracle_rng_seed(seed, int(i)) for i in idx], dtype=np.uint64)
    tot = 0
    for p in plies_list:
        gogame.batch_rollout(st, rng, p, True)
        want, wr, _ = c_oracle.batch_rollout_mt(want, wr, p, True)
        tot += p
        got = st[torch.as_tensor(idx, device='cuda')].cpu().numpy()
        if show(N, got, want, 'bytes N=%d B=%d after %d plies' % (N, B, tot)):
            return False
    return True

ok = True
for N, B in ((19, 8), (19, 64), (9, 40), (13, 33), (5, 16)):
    ok = tracked_run(N, B, [1, 1, 1, 2, 5, 20, 100, 300]) and ok
    if not ok: break
if ok:
    for N, B in ((19, 8192), (19, 65536), (9, 16384), (13, 12000)):
        ok = byte_run(N, B, [2, 3, 30, 100, 300]) and ok
        if not ok: break
print('ALL OK' if ok else 'FAILED')
