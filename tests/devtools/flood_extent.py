"""Developer aid (CPU only): how far do the floods of a fused-rollout ply reach, in ROWS from their seed?

Stationary-mix 19x19 positions (oracle rollouts, de-synchronised like bench.py); for every ply the groups the multi-ply
kernel floods - the mover's group G through the new stone q (when q has a friendly neighbour) and the opponent group at each
neighbour of q - are labelled on the position with the new stone, and their extent above / below the SEED row is recorded.
Printed: the distribution per flood, and per 64-lane batch (16 boards) the largest reach - what a row WINDOW around the
seed row would have to cover for the whole batch to close inside it.
    python tests/devtools/flood_extent.py [boards] [plies]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scipy import ndimage
from oracle import c_oracle

N = 19


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    plies = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    st = np.zeros((B, 6, N, N), np.uint8)
    rng = c_oracle.rng_seed(11, B)
    chunk = max(1, B // 16)
    for sl in range(1, 16):
        lo, hi = sl * chunk, min(B, (sl + 1) * chunk)
        s2, r2, _ = c_oracle.batch_rollout(st[lo:hi], rng[lo:hi], sl * 40, True)
        st[lo:hi], rng[lo:hi] = s2, r2
    st, rng, _ = c_oracle.batch_rollout_mt(st, rng, 1200, True)
    print('mean stones per board: %.1f' % (st[:, :2].sum() / B))
    reach = []          # per flood: max(rows above seed, rows below seed)
    span = []           # per flood: rows spanned
    size = []
    per_batch = []      # per (ply, 16 boards): the largest reach
    nflood = []
    for t in range(plies):
        prev = st
        st, rng, last = c_oracle.batch_rollout_mt(prev, rng, 1, True)
        batch_reach = np.zeros(B, np.int32)
        for b in range(B):
            a = int(last[b])
            if a < 0 or a >= N * N:
                continue
            if prev[b, 5].any():      # the game was reset before this move: the position is the empty board + q
                continue
            turn = int(prev[b, 2, 0, 0])
            r, c = divmod(a, N)
            mine = prev[b, turn].copy(); mine[r, c] = 1
            opp = prev[b, 1 - turn]
            lm, _ = ndimage.label(mine)
            lo_, _ = ndimage.label(opp)
            k = 0
            seen = set()
            friendly = False
            for dr, dc in ((-1, 0), (1, 0), (0, -1), (0, 1)):
                rr, cc = r + dr, c + dc
                if not (0 <= rr < N and 0 <= cc < N):
                    continue
                if mine[rr, cc]:
                    friendly = True
                elif opp[rr, cc]:
                    g = lo_ == lo_[rr, cc]
                    rows = np.flatnonzero(g.any(axis=1))
                    rc = max(rr - rows[0], rows[-1] - rr)
                    reach.append(rc); span.append(rows[-1] - rows[0] + 1); size.append(int(g.sum()))
                    batch_reach[b] = max(batch_reach[b], rc)
                    k += 1
            if friendly:
                g = lm == lm[r, c]
                rows = np.flatnonzero(g.any(axis=1))
                rc = max(r - rows[0], rows[-1] - r)
                reach.append(rc); span.append(rows[-1] - rows[0] + 1); size.append(int(g.sum()))
                batch_reach[b] = max(batch_reach[b], rc)
                k += 1
            nflood.append(k)
        per_batch.extend(batch_reach.reshape(-1, 16).max(axis=1).tolist())
    reach, span, size, per_batch = map(np.array, (reach, span, size, per_batch))
    print('floods per moving board: %.2f' % np.mean(nflood))
    print('group size: mean %.1f, median %d, 90%% %d, 99%% %d, max %d' % (size.mean(), np.median(size), np.percentile(size, 90), np.percentile(size, 99), size.max()))
    print('reach from the seed row (rows), per flood: ' + ' '.join('%d:%.3f' % (k, (reach <= k).mean()) for k in range(0, 19)))
    print('rows spanned, per flood:                  ' + ' '.join('%d:%.3f' % (k, (span <= k).mean()) for k in range(1, 20)))
    print('largest reach per 16-board batch (cdf):    ' + ' '.join('%d:%.3f' % (k, (per_batch <= k).mean()) for k in range(0, 19)))


if __name__ == '__main__':
    main()
