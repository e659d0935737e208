"""Developer aid (CPU only): row sweeps of the FROM-SCRATCH liberty analysis (k_invalid_mask16 / k_track16 / the first classes of a
byte-plane loop) per group of sixteen boards, for two schedules:

  A  the shipped one: six passes, lane (board, half, colour) floods class 2 j + h of the 11-bit constant-weight code from the
     stones next to an empty point of that class, over ALL stones of its colour;
  B  a pre-filter: ONE flood per (board, colour) from the stones with >= 2 empty neighbours (their groups have >= 2 liberties
     whatever else), then the six code passes over the stones that flood did NOT reach (groups whose every stone has <= 1
     empty neighbour: few and small).

A pass lasts as long as the slowest of its 64 lanes (closure tests after sweep 3, 4, 5, ... as flood2_serial_regs does).
Boards: seeded uniform-random 19x19 games with auto-reset, de-synchronised like bench.py (the stationary mix).
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import c_oracle

N = 19
FULL = np.uint32((1 << N) - 1)


def rows(planes):   # [L, N, N] 0/1 -> [L, N] uint32
    w = (1 << np.arange(N, dtype=np.uint32))
    return (planes.astype(np.uint32) * w).sum(axis=2).astype(np.uint32)


def run_fill(s, m):
    s = s & m
    for _ in range(N):
        g = (s | (s << np.uint32(1)) | (s >> np.uint32(1))) & m
        if (g == s).all():
            break
        s = g
    return s


def flood_sweeps(f0, m, first_test=3):
    """f0, m: [L, N].  Gauss-Seidel sweeps down, up, down, ...; returns (fill, sweeps used by the lock-step batch)."""
    f = f0.copy() & m
    # the true fixed point, to know when each sweep count suffices
    k = 0
    while True:
        k += 1
        if k & 1:
            f[:, 0] = run_fill(f[:, 0], m[:, 0])
            for r in range(1, N):
                f[:, r] = run_fill(f[:, r] | (f[:, r - 1] & m[:, r]), m[:, r])
        else:
            f[:, N - 1] = run_fill(f[:, N - 1], m[:, N - 1])
            for r in range(N - 2, -1, -1):
                f[:, r] = run_fill(f[:, r] | (f[:, r + 1] & m[:, r]), m[:, r])
        if k >= first_test:
            # closure: no filled stone next to an unfilled stone vertically (horizontal runs are complete)
            up = np.zeros_like(f); up[:, 1:] = f[:, :-1]
            dn = np.zeros_like(f); dn[:, :-1] = f[:, 1:]
            if not (((up | dn) & m & ~f).any()):
                return f, k


def dil(x):
    d = (x | (x << np.uint32(1)) | (x >> np.uint32(1))) & FULL
    d[:, 1:] |= x[:, :-1]
    d[:, :-1] |= x[:, 1:]
    return d


def codes():
    out = []
    w = 0
    while len(out) < N * N:
        if bin(w).count('1') == 5:
            out.append(w)
        w += 1
    return np.array(out, dtype=np.uint32).reshape(N, N)


def main():
    waves = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    NB = 16
    B = waves * NB
    st = np.zeros((B, 6, N, N), np.uint8)
    rng = c_oracle.rng_seed(11, B)
    chunk = max(1, B // 16)
    for sl in range(16):
        lo, hi = sl * chunk, min(B, (sl + 1) * chunk)
        s2, r2, _ = c_oracle.batch_rollout(st[lo:hi], rng[lo:hi], 640 + sl * 700 // 16, True)
        st[lo:hi], rng[lo:hi] = s2, r2
    perm = np.random.default_rng(5).permutation(B)
    st = st[perm]
    cw = codes()
    cls = [rows(((cw >> k) & 1)[None].astype(np.uint8))[0] for k in range(11)] + [np.zeros(N, np.uint32)]
    totA = totB = 0
    statA, statB1, statB2, ufrac = [], [], [], []
    bad = 0
    for w in range(waves):
        s = st[w * NB:(w + 1) * NB]
        bk, wh = rows(s[:, 0]), rows(s[:, 1])
        e = FULL & ~(bk | wh)
        m = np.concatenate([bk, wh])              # lanes: colour-major, [2 NB, N]
        e2 = np.concatenate([e, e])
        # A
        cnt = np.zeros((2 * NB, N, 32), np.uint8)
        count = np.zeros_like(m, dtype=np.uint32)
        nA = 0
        countA = [np.zeros_like(m) for _ in range(11)]
        for j in range(6):
            mm = np.concatenate([m, m]); ee = np.concatenate([e2 & cls[2 * j][None], e2 & cls[2 * j + 1][None]])
            f, k = flood_sweeps(dil(ee) & mm, mm)
            nA += k
            countA[2 * j] = f[:2 * NB]
            if 2 * j + 1 < 11:
                countA[2 * j + 1] = f[2 * NB:]
        def popc_ge6(fl):
            tot = np.zeros(fl[0].shape + (N,), np.int32)
            for x in fl:
                tot += ((x[..., None] >> np.arange(N, dtype=np.uint32)) & 1).astype(np.int32)
            return tot >= 6
        multiA = popc_ge6(countA)
        # B: pre-filter
        nb_e = [np.zeros_like(m) for _ in range(4)]
        nb_e[0] = (e2 << np.uint32(1)) & FULL; nb_e[1] = e2 >> np.uint32(1)
        nb_e[2][:, 1:] = e2[:, :-1]; nb_e[3][:, :-1] = e2[:, 1:]
        # >= 2 of the four
        a, b, c, d = nb_e
        two = (a & b) | (c & d) | ((a | b) & (c | d))
        R1, k1 = flood_sweeps(two & m, m)
        U = m & ~R1
        nB = k1
        countB = [np.zeros_like(m) for _ in range(11)]
        k2s = []
        for j in range(6):
            mm = np.concatenate([U, U]); ee = np.concatenate([e2 & cls[2 * j][None], e2 & cls[2 * j + 1][None]])
            f, k = flood_sweeps(dil(ee) & mm, mm, first_test=int(os.environ.get('FT', '2')))
            nB += k
            k2s.append(k)
            countB[2 * j] = f[:2 * NB]
            if 2 * j + 1 < 11:
                countB[2 * j + 1] = f[2 * NB:]
        multiB = popc_ge6(countB) | (((R1[..., None] >> np.arange(N, dtype=np.uint32)) & 1) > 0)
        if (multiA != multiB).any():
            bad += 1
        statA.append(nA); statB1.append(k1); statB2.append(sum(k2s))
        ufrac.append(float(np.array([bin(int(x)).count('1') for x in U.ravel()]).sum()) /
                     max(1, np.array([bin(int(x)).count('1') for x in m.ravel()]).sum()))
    print('groups of 16 boards: %d, mismatching groups (B vs A): %d' % (waves, bad))
    print('A: sweeps per group, six passes: mean %.1f (per pass %.2f)' % (np.mean(statA), np.mean(statA) / 6))
    print('B: pre-filter flood %.2f sweeps + six passes over the rest %.1f (per pass %.2f) = %.1f;  stones left to the code passes: %.1f %%'
          % (np.mean(statB1), np.mean(statB2), np.mean(statB2) / 6, np.mean(statB1) + np.mean(statB2), 100 * np.mean(ufrac)))


if __name__ == '__main__':
    main()
