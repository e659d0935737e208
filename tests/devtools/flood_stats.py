"""Developer aid (CPU only): how many row sweeps do the five floods of a fused-rollout ply need, per 12-board wave?

Plays seeded uniform-random 19x19 games with the oracle (de-synchronised like bench.py), and for every ply replays the
flood schedule of k_rollout3 phase 2 (down sweep, up sweep, ... with full run fill per visited row) on the mover's group
from the new stone and the opponent groups at its four neighbours.  Prints the distribution of the sweep index after
which a wave's slowest flood is closed - the number that decides where the closure tests belong.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import c_oracle

N = 19
NB = int(os.environ.get("NB", "16"))


def rows_of(plane):
    w = (1 << np.arange(N, dtype=np.int64))
    return [int(x) for x in (plane.astype(np.int64) * w).sum(axis=1)]


def run_fill(s, m):
    # all bits of m connected to s through runs of ones (both directions)
    while True:
        g = (s | (s << 1) | (s >> 1)) & m
        if g == s:
            return s
        s = g


def full_fill(f, m):
    f = list(f)
    ch = True
    while ch:
        ch = False
        for r in range(N):
            s = f[r]
            if r > 0: s |= f[r - 1] & m[r]
            if r < N - 1: s |= f[r + 1] & m[r]
            s = run_fill(s, m[r])
            if s != f[r]:
                f[r] = s; ch = True
    return f


def sweeps_needed(m, sr, sbit):
    if sr < 0 or sr >= N or not (m[sr] & sbit):
        return 0
    f = [0] * N
    f[sr] = sbit
    target = full_fill(f, m)
    f[sr] = sbit
    k = 0
    while True:
        k += 1
        if k & 1:   # down
            f[0] = run_fill(f[0], m[0])
            for r in range(1, N):
                f[r] = run_fill(f[r] | (f[r - 1] & m[r]), m[r])
        else:
            f[N - 1] = run_fill(f[N - 1], m[N - 1])
            for r in range(N - 2, -1, -1):
                f[r] = run_fill(f[r] | (f[r + 1] & m[r]), m[r])
        if f == target:
            return k


def main():
    waves = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    plies = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    B = waves * NB
    st = np.zeros((B, 6, N, N), np.uint8)
    rng = c_oracle.rng_seed(11, B)
    # de-synchronise: game g has played (g * 700 / B) plies already
    chunk = max(1, B // 16)
    for sl in range(1, 16):
        lo, hi = sl * chunk, min(B, (sl + 1) * chunk)
        if lo < hi:
            s2, r2, _ = c_oracle.batch_rollout(st[lo:hi], rng[lo:hi], sl * 700 // 16, True)
            st[lo:hi], rng[lo:hi] = s2, r2
    hist_wave = np.zeros(64, np.int64)
    hist_flood = np.zeros(64, np.int64)
    for t in range(plies):
        nxt, r2, last = c_oracle.batch_rollout(st, rng, 1, True)
        need = np.zeros(B, np.int64)
        for b in range(B):
            a = int(last[b])
            if a < 0 or a >= N * N:
                continue
            s = st[b]
            if s[5].any():     # finished game: the ply was an auto-reset + first move on an empty board
                s = np.zeros_like(s)
            turn = int(s[2, 0, 0])
            ar, ac = divmod(a, N)
            mine = rows_of(s[turn]); opp = rows_of(s[1 - turn])
            mine[ar] |= 1 << ac
            worst = sweeps_needed(mine, ar, 1 << ac)
            hist_flood[worst] += 1
            for (rr, cc) in ((ar - 1, ac), (ar + 1, ac), (ar, ac - 1), (ar, ac + 1)):
                if 0 <= cc < N:
                    k = sweeps_needed(opp, rr, 1 << cc)
                    hist_flood[k] += 1
                    worst = max(worst, k)
            need[b] = worst
        for w in range(waves):
            hist_wave[need[w * NB:(w + 1) * NB].max()] += 1
        st, rng = nxt, r2
    tot = hist_wave.sum()
    print('per-flood sweeps-to-closure:', {k: int(v) for k, v in enumerate(hist_flood) if v})
    print('per-wave (max of 60 floods):', {k: int(v) for k, v in enumerate(hist_wave) if v})
    mean = (hist_wave * np.arange(64)).sum() / tot
    print('mean per wave %.2f' % mean)
    for first_test in (1, 2, 3):
        cost = 0.0
        for k, v in enumerate(hist_wave):
            sw = max(k, first_test)
            cost += v * (266 * sw + 130 * (sw - first_test + 1))
        print('first closure test after sweep %d: %.0f cycles per ply per wave' % (first_test, cost / tot))


if __name__ == '__main__':
    main()
