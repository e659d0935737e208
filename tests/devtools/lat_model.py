"""Lane-by-lane NumPy model of gymgo_amd/csrc/gg_lat.h (k_rollout_lat): the SAME operations on 64-entry uint32 vectors
(DPP moves as index shuffles, v_bitop3 as its truth table), checked against the C oracle on the CPU.  A development aid:
it validates the bit tricks (field packing, alternating-order run fills, 8-bit board sums) and the class patch before a
GPU box sees the kernel.  Usage: python tests/devtools/lat_model.py [N] [waves] [plies]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import c_oracle  # noqa: E402

U32 = np.uint32
LANES = np.arange(64)


def u32(x):
    return np.asarray(x).astype(np.uint64).astype(U32) if not isinstance(x, np.ndarray) or x.dtype != U32 else x


def B3(a, b, c, t):
    a, b, c = u32(a), u32(b), u32(c)
    out = np.zeros(64, U32)
    for i in range(8):
        if (t >> i) & 1:
            ta = a if (i >> 2) & 1 else ~a
            tb = b if (i >> 1) & 1 else ~b
            tc = c if i & 1 else ~c
            out |= ta & tb & tc
    return out


TA, TB, TC = 0xF0, 0xCC, 0xAA
T_ANDOR = (TA & TB) | TC
T_SEL = (TA & TB) | (~TA & TC & 0xFF)
T_AND_ANDN = TA & TB & (~TC & 0xFF)
T_OR3 = TA | TB | TC


def brev(x):
    x = u32(x).copy()
    x = ((x >> U32(1)) & U32(0x55555555)) | ((x & U32(0x55555555)) << U32(1))
    x = ((x >> U32(2)) & U32(0x33333333)) | ((x & U32(0x33333333)) << U32(2))
    x = ((x >> U32(4)) & U32(0x0F0F0F0F)) | ((x & U32(0x0F0F0F0F)) << U32(4))
    x = ((x >> U32(8)) & U32(0x00FF00FF)) | ((x & U32(0x00FF00FF)) << U32(8))
    return (x >> U32(16)) | (x << U32(16))


def popc(x):
    x = u32(x)
    return np.array([bin(int(v)).count('1') for v in x], U32)


class Model:
    def __init__(self, R):
        self.R = R
        self.LPB = 16 if R <= 15 else 32
        self.NBW = 64 // self.LPB
        self.FW = 10 if R <= 9 else (16 if R <= 15 else 32)
        self.NF = 32 // self.FW
        self.NREG = (5 + self.NF - 1) // self.NF      # the five-flood fallback
        self.NREG3 = (3 + self.NF - 1) // self.NF     # the fast path: two opponent slots + G
        self.fallbacks = 0
        self.plays = 0
        self.FM = 0xFFFFFFFF if self.FW >= 32 else (1 << self.FW) - 1
        self.kBits = 16 if R <= 16 else 32
        self.kSat = R > 9

    # ---- DPP
    def above(self, x):     # lane i reads lane i - 1
        out = np.zeros(64, U32)
        out[1:] = x[:-1]
        if self.LPB == 16:
            out[LANES % 16 == 0] = 0
        return out

    def below(self, x):     # lane i reads lane i + 1
        out = np.zeros(64, U32)
        out[:-1] = x[1:]
        if self.LPB == 16:
            out[LANES % 16 == 15] = 0
        return out

    def row_shr(self, x, n):
        out = np.zeros(64, U32)
        for i in range(64):
            if (i % 16) - n >= 0:
                out[i] = x[i - n]
        return out

    def row_ror(self, x, n):
        out = np.zeros(64, U32)
        for i in range(64):
            out[i] = x[(i // 16) * 16 + ((i % 16) - n) % 16]
        return out

    def dilate(self, x):
        return B3(x + x, x >> U32(1), self.above(x), T_OR3) | self.below(x)

    def board_sum(self, x):
        x = u32(x)
        for n in (8, 4, 2, 1):
            x = x + self.row_ror(x, n)
        if self.LPB == 32:
            x = x + x[LANES ^ 16]
        return x

    def board_max(self, x):
        x = np.asarray(x, np.int64)
        for n in (8, 4, 2, 1):
            y = np.array([x[(i // 16) * 16 + ((i % 16) - n) % 16] for i in range(64)])
            x = np.maximum(x, y)
        if self.LPB == 32:
            x = np.maximum(x, x[LANES ^ 16])
        return x

    def board_scan(self, v):
        v = u32(v)
        for n in (1, 2, 4, 8):
            v = v + self.row_shr(v, n)
        if self.LPB == 32:
            add = np.zeros(64, U32)
            for i in range(64):
                if (i // 16) in (1, 3):
                    add[i] = v[(i // 16) * 16 - 1]
            v = v + add
        return v

    def kth_bit(self, v, tt):
        v, tt = u32(v), u32(tt).copy()
        ps = np.zeros(64, U32)
        sh = self.kBits // 2
        while sh >= 1:
            c = popc((v >> ps) & U32((1 << sh) - 1))
            d = tt - c
            lt = (d.astype(np.int32) >> 31).astype(U32)
            tt = B3(lt, tt, d, T_SEL)
            ps = B3(ps, np.full(64, sh, U32), lt, TA | (TB & ~TC & 0xFF))
            sh >>= 1
        return ps

    @staticmethod
    def visit(ma, mb, s):
        t = ma + s
        u = B3(t, s, ma, T_SEL)
        v = brev(u)
        t2 = mb + v
        return B3(t2, v, mb, T_SEL)

    def flood(self, F, Mk, Mkr):
        K = len(F)
        F = [self.visit(Mk[k], Mkr[k], F[k]) for k in range(K)]
        for it in range(512):
            o = [B3(self.above(F[k]) | self.below(F[k]), Mkr[k], F[k], T_AND_ANDN) for k in range(K)]
            if not any(x.any() for x in o):
                return [brev(F[k]) for k in range(K)]
            s = [o[k] | F[k] for k in range(K)]
            s = [B3(self.above(s[k]) | self.below(s[k]), Mkr[k], s[k], T_ANDOR) for k in range(K)]   # the second row up / down
            F = [self.visit(Mkr[k], Mk[k], s[k]) for k in range(K)]
            o = [B3(self.above(F[k]) | self.below(F[k]), Mk[k], F[k], T_AND_ANDN) for k in range(K)]
            if not any(x.any() for x in o):
                return F
            s = [o[k] | F[k] for k in range(K)]
            s = [B3(self.above(s[k]) | self.below(s[k]), Mk[k], s[k], T_ANDOR) for k in range(K)]
            F = [self.visit(Mk[k], Mkr[k], s[k]) for k in range(K)]
        raise RuntimeError('flood bound')

    # ---- the constant-weight code table (gg_v2.h make_cw_table)
    _cw = None

    @classmethod
    def cw(cls):
        if cls._cw is None:
            t = np.zeros((12, 20), U32)
            q = 0
            for w in range(1 << 11):
                if q >= 361:
                    break
                if bin(w).count('1') != 5:
                    continue
                r, c = divmod(q, 19)
                for i in range(11):
                    if (w >> i) & 1:
                        t[i][r] |= U32(1 << c)
                q += 1
            cls._cw = t
        return cls._cw

    @staticmethod
    def classify11(w):
        def cs(a, b, c): return a ^ b ^ c
        def cc(a, b, c): return (a & b) | (c & (a | b))
        s0, c0 = cs(w[0], w[1], w[2]), cc(w[0], w[1], w[2])
        s1, c1 = cs(w[3], w[4], w[5]), cc(w[3], w[4], w[5])
        s2, c2 = cs(w[6], w[7], w[8]), cc(w[6], w[7], w[8])
        s3, c3 = w[9] ^ w[10], w[9] & w[10]
        ss, cs_ = cs(s0, s1, s2), cc(s0, s1, s2)
        t = ss & s3
        u0, v0 = cs(c0, c1, c2), cc(c0, c1, c2)
        u1, v1 = cs(c3, cs_, t), cc(c3, cs_, t)
        bit1, v2 = u0 ^ u1, u0 & u1
        bit2, bit3 = cs(v0, v1, v2), cc(v0, v1, v2)
        return bit3 | (bit2 & bit1)

    def classes(self, bl, wh, full, r):
        E = full & ~(bl | wh)
        cw = self.cw()
        d = [self.dilate(E & cw[i][np.minimum(r, 19)]) for i in range(11)]
        NC = 1 if self.NF >= 2 else 2
        multi_all = np.zeros(64, U32)
        for c in range(NC):
            P = (bl | (wh << U32(self.FW & 31))) if NC == 1 else (wh if c else bl)
            Pr = brev(P)
            F = [((d[i] | (d[i] << U32(self.FW & 31))) if NC == 1 else d[i]) & P for i in range(11)]
            F = self.flood(F, [P] * 11, [Pr] * 11)
            multi = self.classify11(F)
            multi_all |= ((multi & U32(self.FM)) | (multi >> U32(self.FW & 31))) if NC == 1 else multi
        return multi_all & full

    # ---- one wave
    def run(self, states, rng, N, plies, auto_reset=True, check=None):
        """states uint8 [nb, 6, N, N] (nb <= NBW), rng uint64 [nb]; returns new states, rng, last actions."""
        nb = states.shape[0]
        LPB, NF, FW, NREG = self.LPB, self.NF, self.FW, self.NREG
        r = LANES % LPB
        j = LANES // LPB
        on = j < nb
        full = np.where(r < N, U32((1 << N) - 1), U32(0)).astype(U32)
        P = N * N

        def rows(plane_idx):
            out = np.zeros(64, U32)
            for l in range(64):
                if on[l] and r[l] < N:
                    bits = states[j[l], plane_idx, r[l]]
                    out[l] = sum(int(v) << c for c, v in enumerate(bits))
            return out
        bl, wh, inv = rows(0), rows(1), rows(3)
        fl = np.zeros(64, U32)
        for l in range(64):
            if on[l]:
                s = states[j[l]]
                fl[l] = (1 if s[2, 0, 0] else 0) | (2 if s[4, 0, 0] else 0) | (4 if s[5, 0, 0] else 0)
        M = self.classes(bl, wh, full, r)
        turn = (fl & U32(1)) != 0
        me = np.where(turn, wh, bl).astype(U32)
        op = np.where(turn, bl, wh).astype(U32)
        x = np.array([rng[min(jj, nb - 1)] for jj in j], np.uint64)
        lastv = np.full(64, -1, np.int64)
        played = np.zeros(64, np.int64)
        for t in range(plies):
            done = (fl & U32(4)) != 0
            live = on & ~(done & (not auto_reset))
            if not live.any():
                break
            rs = live & done
            if rs.any():
                keep = np.where(rs, U32(0), U32(0xFFFFFFFF)).astype(U32)
                me, op, M, inv, fl = me & keep, op & keep, M & keep, inv & keep, fl & keep
            lv = np.where(live, U32(0xFFFFFFFF), U32(0)).astype(U32)
            valid = full & ~inv
            cnt = popc(valid)
            incl = self.board_scan(cnt)
            total = self.board_sum(cnt)
            with np.errstate(over='ignore'):
                xn = x + np.uint64(0x9E3779B97F4A7C15)
                z = xn.copy()
                z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
                z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
                u = z ^ (z >> np.uint64(31))
            x = np.where(live, xn, x)
            k = (((u >> np.uint64(32)) * (total.astype(np.uint64) + np.uint64(1))) >> np.uint64(32)).astype(U32)
            tt = k - (incl - cnt)
            hit = live & (tt < cnt)
            pos = self.kth_bit(valid, tt)
            Q = np.where(hit, U32(1) << pos, U32(0)).astype(U32)
            pas = k == total
            lastv = np.where(live, np.where(hit, r * N + pos.astype(np.int64), np.where(pas, P, -1)), lastv)
            played += live
            me, op, M, inv, fl = self.play(me, op, M, inv, fl, Q, pas, lv, full)
            if check is not None:
                check(t, self._emit(me, op, inv, fl, N, nb, r, j, on), M, full)
        out = self._emit(me, op, inv, fl, N, nb, r, j, on)
        lastb = self.board_max(lastv)
        return out, np.array([x[jj * LPB] for jj in range(nb)], np.uint64), np.array([lastb[jj * LPB] for jj in range(nb)]), \
            np.array([played[jj * LPB] for jj in range(nb)])

    def play_full(self, me, op, M, inv, fl, Q, pas, lv, full):
        """the five-flood ply (round 5): the fallback of play() when q touches three distinct opponent groups with >= 2 liberties"""
        me1 = me | Q
        su, sd = self.below(Q), self.above(Q)
        sl, sr = Q >> U32(1), Q + Q
        opn = B3(B3(su, sd, sl, T_OR3) | sr, full, op, T_AND_ANDN)
        seeds = [su & op, sd & op, sl & op, sr & op, Q]
        F = [np.zeros(64, U32) for _ in range(self.NREG)]
        Mk = [np.zeros(64, U32) for _ in range(self.NREG)]
        for f in range(5):
            sh = U32((self.FW * (f % self.NF)) & 31)
            F[f // self.NF] |= seeds[f] << sh
            Mk[f // self.NF] |= (op if f < 4 else me1) << sh
        Mkr = [brev(m) for m in Mk]
        F = self.flood(F, Mk, Mkr)
        fr = [F[f] if self.NF == 1 else (F[f // self.NF] >> U32((self.FW * (f % self.NF)) & 31)) & U32(self.FM) for f in range(5)]
        U = B3(fr[0], fr[1], fr[2], T_OR3) | fr[3]
        G = fr[4]
        C = U & ~M
        E1 = full & ~(me1 | op)
        EG = E1 | C
        Ee = [np.zeros(64, U32) for _ in range(self.NREG)]
        for f in range(5):
            Ee[f // self.NF] |= (E1 if f < 4 else EG) << U32((self.FW * (f % self.NF)) & 31)
        Lb = [self.dilate(F[k2]) & Ee[k2] for k2 in range(self.NREG)]
        W1 = np.zeros(64, U32)
        W2 = np.zeros(64, U32)
        for f in range(5):
            c = popc(Lb[f] if self.NF == 1 else Lb[f // self.NF] & U32((self.FM << ((self.FW * (f % self.NF)) & 31)) & 0xFFFFFFFF))
            if self.kSat:
                c = np.minimum(c, U32(2))
            if f < 4:
                W1 |= c << U32(8 * f)
            else:
                W2 = c.copy()
        pc = np.minimum(popc(C), U32(2))
        W2 |= (pc << U32(8)) | np.where(opn != 0, U32(0x10000), U32(0)).astype(U32)
        S1, S2 = self.board_sum(W1), self.board_sum(W2)
        g1 = S1 + U32(0x7E7E7E7E)
        M1 = B3(M, U, G, TA & ~(TB | TC) & 0xFF)
        for f in range(4):
            ge = ((g1 << U32(24 - 8 * f)).astype(np.int32) >> 31).astype(U32)
            M1 = B3(ge, fr[f], M1, T_ANDOR)
        geG = ((U32(1) - (S2 & U32(0xFF))).astype(np.int32) >> 31).astype(U32)
        M1 = B3(geG, G, M1, T_ANDOR)
        K = np.zeros(64, U32)
        if (C != 0).any():
            kc = (((S2 >> U32(8)) & U32(0xFF)) == 1) & (((S2 >> U32(16)) & U32(0xFF)) == 0)
            K = np.where(kc, C, U32(0)).astype(U32)
            Xm = B3(me1, M, G, TA & ~(TB | TC) & 0xFF)
            X = self.dilate(C) & Xm
            if (X != 0).any():
                X = self.flood([X], [Xm], [brev(Xm)])[0]
                M1 = M1 | X
        op2 = op & ~C
        E2 = full & ~(me1 | op2)
        xs = E2 | B3(M1, op2, me1, T_SEL)
        nbs = self.dilate(xs)
        inv2 = B3(full, E2, nbs, TA & ~(TB & TC) & 0xFF) | K
        me = B3(lv, op2, me1, T_SEL)
        op = B3(lv, me1, op2, T_SEL)
        inv = B3(lv, inv2, inv, T_SEL)
        M = M1
        pm = np.where(pas, U32(0xFFFFFFFF), U32(0)).astype(U32)
        fl2 = ((fl ^ U32(1)) & U32(1)) | (pm & U32(2)) | (pm & (fl << U32(1)) & U32(4))
        fl = B3(lv, fl2, fl, T_SEL)
        return me, op, M, inv, fl

    def play(self, me, op, M, inv, fl, Q, pas, lv, full):
        """lat_play (round 6): at most TWO distinct opponent groups next to q keep their class question open (the ones in M: an
        opponent group NOT in M had q as its only liberty and is captured whatever its shape), so three floods do: slot A = every
        atari neighbour + the FIRST neighbour in M (priority up, down, left, right), slot B = the SECOND one, G = the mover's
        group through q.  Further neighbours in M must be covered by A | B (the same groups); if not (1.2 % of the moves) a
        second round floods the left-over (left / right) neighbours in slots of their own."""
        NF, FW = self.NF, self.FW
        self.plays += 1
        me1 = me | Q
        su, sd = self.below(Q), self.above(Q)
        sl, sr = Q >> U32(1), Q + Q
        nbr = B3(su, sd, sl, T_OR3) | sr
        opn = B3(nbr, full, op, T_AND_ANDN)
        nbo = nbr & op
        mAll = nbo & M
        aAll = nbo & ~M
        mUl, mDl, mL, mR = su & mAll, sd & mAll, sl & mAll, sr & mAll
        fU = self.above(mUl)          # lane r: the point above q is an opponent stone in M (bit c)
        fD = self.below(mDl)
        gU = self.above(fU)           # lane r + 1 learns about the point above q
        fL, fR = mL + mL, mR >> U32(1)
        ud = fU | fD
        p1L = fL & ~ud
        p2L = fL & (fU ^ fD)
        p1R = fR & ~(ud | fL)
        one3 = B3(fU, fD, fL, 0x16)
        p2R = fR & one3
        seedA = aAll | mUl | (mDl & ~gU) | (p1L >> U32(1)) | (p1R + p1R)
        seedB = (mDl & gU) | (p2L >> U32(1)) | (p2R + p2R)
        NR = self.NREG3
        seeds = [seedA, seedB, Q]
        F = [np.zeros(64, U32) for _ in range(NR)]
        Mk = [np.zeros(64, U32) for _ in range(NR)]
        for f in range(3):
            sh = U32((FW * (f % NF)) & 31)
            F[f // NF] |= seeds[f] << sh
            Mk[f // NF] |= (op if f < 2 else me1) << sh
        Mkr = [brev(m) for m in Mk]
        F = self.flood(F, Mk, Mkr)
        fr = [F[f] if NF == 1 else (F[f // NF] >> U32((FW * (f % NF)) & 31)) & U32(self.FM) for f in range(3)]
        U = fr[0] | fr[1]
        G = fr[2]
        C = fr[0] & ~M
        E1 = full & ~(me1 | op)
        EG = E1 | C
        Ee = [np.zeros(64, U32) for _ in range(NR)]
        for f in range(3):
            Ee[f // NF] |= (E1 if f < 2 else EG) << U32((FW * (f % NF)) & 31)
        Lb = [self.dilate(F[k2]) & Ee[k2] for k2 in range(NR)]
        W = np.zeros(64, U32)
        for f in range(3):
            c = popc(Lb[f] if NF == 1 else Lb[f // NF] & U32((self.FM << ((FW * (f % NF)) & 31)) & 0xFFFFFFFF))
            if self.kSat:
                c = np.minimum(c, U32(2))
            W |= c << U32(8 * f)
        pc = np.minimum(popc(C), U32(2))
        W |= (pc << U32(24)) | np.where(opn != 0, U32(0x40000000), U32(0)).astype(U32)
        S = self.board_sum(W)
        M1x = np.zeros(64, U32)
        if (mAll & ~U).any():         # a third distinct group in M (only the left / right neighbour can be left over): round two
            self.fallbacks += 1
            s2 = [B3(sl, mAll, U, T_AND_ANDN), B3(sr, mAll, U, T_AND_ANDN)]
            K2 = 2 if NF == 1 else 1
            F2 = [s2[0], s2[1]] if NF == 1 else [s2[0] | (s2[1] << U32(FW & 31))]
            F2 = self.flood(F2, Mk[:K2], Mkr[:K2])
            Eb = E1 if NF == 1 else (E1 | (E1 << U32(FW & 31)))
            Lb2 = [self.dilate(F2[k]) & Eb for k in range(K2)]
            W2 = np.zeros(64, U32)
            f2 = []
            for f in range(2):
                f2.append(F2[f] if NF == 1 else (F2[0] >> U32((FW * f) & 31)) & U32(self.FM))
                c = popc(Lb2[f] if NF == 1 else Lb2[0] & U32((self.FM << ((FW * f) & 31)) & 0xFFFFFFFF))
                if self.kSat:
                    c = np.minimum(c, U32(2))
                W2 |= c << U32(8 * f)
            g2 = self.board_sum(W2) + U32(0x00007E7E)
            for f in range(2):
                ge = ((g2 << U32(24 - 8 * f)).astype(np.int32) >> 31).astype(U32)
                M1x |= ge & f2[f]
            U = U | f2[0] | f2[1]
        g1 = S + U32(0x007E7E7E)
        M1 = B3(M, U, G, TA & ~(TB | TC) & 0xFF) | M1x
        frm = [fr[0] & M, fr[1], G]
        for f in range(3):
            ge = ((g1 << U32(24 - 8 * f)).astype(np.int32) >> 31).astype(U32)
            M1 = B3(ge, frm[f], M1, T_ANDOR)
        K = np.zeros(64, U32)
        if (C != 0).any():
            kc = (((S >> U32(24)) & U32(0x3F)) == 1) & ((S >> U32(30)) == 0)
            K = np.where(kc, C, U32(0)).astype(U32)
            Xm = B3(me1, M, G, TA & ~(TB | TC) & 0xFF)
            X = self.dilate(C) & Xm
            if (X != 0).any():
                X = self.flood([X], [Xm], [brev(Xm)])[0]
                M1 = M1 | X
        op2 = op & ~C
        E2 = full & ~(me1 | op2)
        xs = E2 | B3(M1, op2, me1, T_SEL)
        nbs = self.dilate(xs)
        inv2 = B3(full, E2, nbs, TA & ~(TB & TC) & 0xFF) | K
        me = B3(lv, op2, me1, T_SEL)
        op = B3(lv, me1, op2, T_SEL)
        inv = B3(lv, inv2, inv, T_SEL)
        M = M1
        pm = np.where(pas, U32(0xFFFFFFFF), U32(0)).astype(U32)
        fl2 = ((fl ^ U32(1)) & U32(1)) | (pm & U32(2)) | (pm & (fl << U32(1)) & U32(4))
        fl = B3(lv, fl2, fl, T_SEL)
        return me, op, M, inv, fl

    def _emit(self, me, op, inv, fl, N, nb, r, j, on):
        out = np.zeros((nb, 6, N, N), np.uint8)
        turn = (fl & U32(1)) != 0
        bl = np.where(turn, op, me)
        wh = np.where(turn, me, op)
        for l in range(64):
            if on[l] and r[l] < N:
                b = j[l]
                for c in range(N):
                    out[b, 0, r[l], c] = (int(bl[l]) >> c) & 1
                    out[b, 1, r[l], c] = (int(wh[l]) >> c) & 1
                    out[b, 3, r[l], c] = (int(inv[l]) >> c) & 1
                out[b, 2, r[l], :] = int(fl[l]) & 1
                out[b, 4, r[l], :] = (int(fl[l]) >> 1) & 1
                out[b, 5, r[l], :] = (int(fl[l]) >> 2) & 1
        return out


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    waves = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    plies = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    R = 9 if N <= 9 else 13 if N <= 13 else 19
    m = Model(R)
    bad = 0
    for w in range(waves):
        nb = m.NBW if w % 3 != 2 else max(1, m.NBW - 1)
        states = np.zeros((nb, 6, N, N), np.uint8)
        rng = c_oracle.rng_seed(1234 + w, nb)
        # two launches: the second starts from mid-game boards (first classes from scratch)
        for launch in range(2):
            ref, ref_rng, ref_last = c_oracle.batch_rollout(states, rng, plies, True)
            got, grng, glast, gplayed = m.run(states, rng, N, plies, True)
            ok = np.array_equal(got, ref) and np.array_equal(grng, ref_rng) and np.array_equal(glast, ref_last)
            if not ok:
                bad += 1
                # find the first ply that differs
                for t in range(1, plies + 1):
                    a, _, _ = c_oracle.batch_rollout(states, rng, t, True)
                    g, _, _, _ = m.run(states, rng, N, t, True)
                    if not np.array_equal(a, g):
                        d = np.argwhere(a != g)
                        print('N=%d wave %d launch %d: first mismatch at ply %d, %d cells, first %s' % (N, w, launch, t, len(d), d[:4].tolist()))
                        break
            states, rng = ref, ref_rng
        print('wave %d done (%s)' % (w, 'ok' if not bad else 'MISMATCH'), flush=True)
    print('model vs oracle: N=%d, %d waves x 2 launches x %d plies: %s  (second flood round on %d of %d wave-plies)'
          % (N, waves, plies, 'OK' if not bad else '%d mismatching launches' % bad, m.fallbacks, m.plays))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
