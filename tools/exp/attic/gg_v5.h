// gg_v5.h - per-ply kernels for big batches: THIRTY-TWO BOARDS PER WAVEFRONT, one lane per (board, colour), every
// liberty class from scratch with ALL 64 lanes flooding.
//
// The two-boards-per-wave kernels (gg_v2.h) run the 22 floods of a board (11 liberty classes x 2 colours, constant-weight
// code) side by side: 44 of 64 lanes carry a flood, and every point-wise rule runs in the "row per lane" layout where
// 19 of 32 lanes do work.  Here the floods run CLASS-MAJOR instead: lane (b, c) holds all rows of colour c of board b in
// registers for the whole step, pass k floods class k of all 64 (board, colour) units at once - eleven passes for
// thirty-two boards (0.34 wave-floods per board instead of 0.5, no idle lane) - and the lane counts, in four bit-sliced
// counter planes of its own, how many of the eleven floods reach each of its stones (>= 1: the group has a liberty,
// >= 6: it has two or more; gg_v2.h explains the code).  The class masks of a pass are wave-uniform (scalar operands),
// nothing goes through LDS between the passes, and every other step of gogame.next_state - placing the stone,
// state_utils.update_pieces (gym_go/state_utils.py:159-180), the ko rule (gym_go/gogame.py:72-75),
// state_utils.compute_invalid_moves (:24-83) - runs on the same 19-rows-per-lane registers: vertical neighbours are
// neighbouring registers, the other colour of the board sits in the partner lane (one DPP quad_perm swap per row), so the
// point-wise work of 32 boards costs what two boards cost in the row-per-lane layout.  Boards come in by one aligned
// 16-byte vector stream per lane (its own plane, 361 bytes), bit-packed with v_dot4; they leave through the group
// emitter of the multi-ply kernel (16 boards = one contiguous byte range, aligned 1 KB blocks).
//
// 65 536 boards are 2 048 waves = two per SIMD: the kernel is compiled for that (256 VGPRs, no spills).  Batches that
// leave SIMDs empty at 32 boards per wave stay on the gg_v2.h kernels (dispatch in gg_kernels.hip).
#pragma once
#include "gg_v4.h"

namespace gg {

constexpr int kNB5 = 32;

template <int R>
struct Lds5 {
  static constexpr int RS = Cfg<R>::kRowStride;
  static constexpr int kRows = 0;                                   // [3][32][RS]: black, white, invalid rows of the results
  static constexpr int kFlags = kRows + 3 * kNB5 * RS;              // [32]: bit 0 turn, 1 passed, 2 done
  static constexpr int kGrpBits = (kFlags + kNB5 + 3) & ~3;         // the bit-strings of the two groups of 16 boards ...
  static constexpr int kGrpWords = ((15 + 16 * 6 * R * R + 31) / 32 + 4) & ~3;
  static constexpr int kGrpLut = kGrpBits + 2 * kGrpWords;          // ... and the 8 bits -> 8 bytes table
  static constexpr int kTotal = kGrpLut + 512;
};

constexpr int QP_SWAP = 0xB1;   // quad_perm [1,0,3,2]: the partner lane (the other colour of the board)

template <int R>
__device__ __forceinline__ void dilate_regs(const uint32_t (&x)[R], uint32_t (&d)[R]) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t hz = r > 0 ? B3(shl1(x[r]), x[r] >> 1, x[r - 1], T_OR3) : (shl1(x[r]) | (x[r] >> 1));
    d[r] = r + 1 < R ? (hz | x[r + 1]) : hz;
  }
}

// flood2_serial (gg_common.h) with the converged fill returned in registers, normal bit order
template <int R>
__device__ __forceinline__ void flood2_serial_regs(const uint32_t (&m)[R], const uint32_t (&mrev)[R], uint32_t (&f)[R],
                                                   uint32_t (&res)[R]) {
#pragma unroll
  for (int r = 1; r < R; r += 2) f[r] = __brev(f[r]);
#pragma unroll 1
  for (int it = 0; it < R * R; ++it) {
#pragma unroll
    for (int r = 0; r < R; ++r) FLOOD_VISIT(r, r - 1, (r & 1) != 0);
    if (it > 0) {
      uint32_t open = 0, above = 0;
#pragma unroll
      for (int r = R - 1; r >= 0; --r) {
        const uint32_t g = ((r + 1) & 1) ? __brev(f[r]) : f[r];
        res[r] = g;
        if (r < R - 1) open |= B3(above, m[r], g, T_AND_ANDN);
        above = g;
      }
      if (__ballot(open != 0) == 0) return;
    }
#pragma unroll
    for (int r = R - 1; r >= 0; --r) FLOOD_VISIT(r, r + 1, ((r + 1) & 1) != 0);
    if (it > 0) {
      uint32_t open = 0, below = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t g = (r & 1) ? __brev(f[r]) : f[r];
        res[r] = g;
        if (r > 0) open |= B3(below, m[r], g, T_AND_ANDN);
        below = g;
      }
      if (__ballot(open != 0) == 0) return;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) res[r] = (r & 1) ? __brev(f[r]) : f[r];
}

// One byte plane of an R x R board (R*R bytes of 0/1 at g, any alignment) -> its R row masks, by THIS lane alone: the
// aligned 16-byte vectors that cover the plane are all requested up front (every lane of the wave streams its own plane),
// each is packed to 16 bits with two v_dot4 chains, the bit-string is shifted into place and cut into rows.
template <int R>
__device__ __forceinline__ void load_plane_rows(const uint8_t *g, uint32_t (&m)[R]) {
  constexpr int P = R * R;
  constexpr int NV = (P + 15 + 15) / 16;          // vectors covering the plane at the worst alignment
  constexpr int NW = NV / 2 + 1;
  const uint32_t mis = (uint32_t)((uintptr_t)g & 15u);
  const uint4 *ga = reinterpret_cast<const uint4 *>(g - mis);
  const int last = (int)((mis + P - 1) >> 4);     // the last vector that holds a byte of the plane
  uint4 d[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) d[v] = ga[v < last ? v : last];   // (unconditional loads on a clamped index: all in flight)
  uint32_t w[NW + 1];
#pragma unroll
  for (int i = 0; i <= NW; ++i) w[i] = 0;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    uint32_t lo = __builtin_amdgcn_udot4(d[v].x & 0x01010101u, 0x08040201u, 0u, false);
    lo = __builtin_amdgcn_udot4(d[v].y & 0x01010101u, 0x80402010u, lo, false);
    uint32_t hi = __builtin_amdgcn_udot4(d[v].z & 0x01010101u, 0x08040201u, 0u, false);
    hi = __builtin_amdgcn_udot4(d[v].w & 0x01010101u, 0x80402010u, hi, false);
    const uint32_t b16 = lo | (hi << 8);
    w[v >> 1] |= (v & 1) ? (b16 << 16) : b16;
  }
  // bit mis + q of the string = byte q of the plane
#pragma unroll
  for (int i = 0; i < NW; ++i) w[i] = __builtin_amdgcn_alignbit(w[i + 1], w[i], mis);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    constexpr uint32_t full = (1u << R) - 1u;
    const int q = r * R;
    m[r] = __builtin_amdgcn_alignbit(w[(q >> 5) + 1], w[q >> 5], (uint32_t)(q & 31)) & full;
  }
}

// emit_group (gg_v4.h) in two instalments.  A wave-iteration of this kernel is load -> ~11 flood passes -> store, and with
// two waves per SIMD nothing else hides the store burst at the end.  But five of the six planes of the result are known
// BEFORE the analysis: the turn / pass / game-over planes follow from the flags and the stone planes are the input plus the
// new stone unless the move captures (about one move in nine).  So the group's byte range is written twice:
//   EARLY (before the passes): every aligned 16-byte vector that holds no byte of an invalid-move plane, from the rows
//         as they are then - these stores drain while the floods run;
//   LATE  (after the mask is known): the vectors that overlap an invalid-move plane, every vector of a board in `redo`
//         (its stones changed: a capture), and the ragged bytes at either end of the group.
// Stores of one wave to one address land in program order, so a vector rewritten LATE ends up with the final bytes.
// rows[p * PL + board * RS + r]: p = 0 black, 1 white, 2 invalid (zero in the EARLY call).
// The two steps of an instalment: (1) the bit-string of a group of <= 16 boards (bit i = byte i of the group's range,
// counted from the 16-byte boundary below its first byte) from the row masks in LDS, (2) rounds [r_lo, r_hi) of the
// aligned-vector stores (round t = vectors v0 + 64 t + lane).  The EARLY rounds are interleaved with the flood passes -
// a burst of all of a group's stores at once stalls the wave at issue until the memory system has taken them, so
// nothing would overlap - the LATE rounds run back to back.
template <int R, bool LATE>
__device__ __forceinline__ void emit_build(const uint8_t *g, int nbrd, const uint32_t *rows, int PL, int RS, const uint32_t *flagsv,
                                           uint32_t *bs, int lane) {
  constexpr int N = R, P = R * R, S = 6 * P, RPL = (R + 3) / 4;
  const uint32_t mo = (uint32_t)((uintptr_t)g & 15u);
  const int nbits = (int)mo + nbrd * S;
  const int q4 = lane >> 2, r04 = RPL * (lane & 3);
  for (int i = lane; i < (nbits + 31) / 32 + 1; i += kWave) bs[i] = 0;
  WAVE_SYNC();
  if (q4 < nbrd) {
    const uint32_t fl = flagsv[q4];
    constexpr uint32_t fullrow = (1u << N) - 1u;
    const uint32_t tp = (fl & 1u) ? fullrow : 0u, pp = (fl & 2u) ? fullrow : 0u, dp = (fl & 4u) ? fullrow : 0u;
    const uint32_t base = mo + (uint32_t)(q4 * S);
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const int rr = r04 + r;
      if (rr < N) {
        const uint32_t rw[6] = {rows[0 * PL + q4 * RS + rr], rows[1 * PL + q4 * RS + rr], tp, LATE ? rows[2 * PL + q4 * RS + rr] : 0u, pp, dp};
#pragma unroll
        for (int p = 0; p < 6; ++p) {
          if (rw[p]) {
            const uint32_t q = base + (uint32_t)(p * P + rr * N);
            const uint64_t x = (uint64_t)rw[p] << (q & 31u);
            atomicOr(bs + (q >> 5), (uint32_t)x);
            if ((uint32_t)(x >> 32)) atomicOr(bs + (q >> 5) + 1, (uint32_t)(x >> 32));
          }
        }
      }
    }
  }
  WAVE_SYNC();
}

template <int R>
__host__ __device__ constexpr int emit_rounds() { return (15 + 16 * 6 * R * R) / 16 / kWave + 1; }   // covers any alignment

template <int R, bool LATE>
__device__ __forceinline__ void emit_store_rounds(uint8_t *g, int nbrd, const uint32_t *bs, const uint2 *lut, uint32_t redo,
                                                  int lane, int r_lo, int r_hi) {
  constexpr int P = R * R, S = 6 * P;
  constexpr uint32_t invS = ((1u << 20) + S - 1) / S;
  const uint32_t mo = (uint32_t)((uintptr_t)g & 15u);
  const int nbits = (int)mo + nbrd * S;
  uint8_t *ga = g - mo;
  const int v0 = mo ? 1 : 0, v1 = nbits >> 4;
  const uint8_t *bb = reinterpret_cast<const uint8_t *>(bs);
  typedef uint32_t u4v __attribute__((ext_vector_type(4)));
  u4v *gv = reinterpret_cast<u4v *>(__builtin_assume_aligned(ga, 16));
  for (int t = r_lo; t < r_hi; ++t) {
    const int v = v0 + kWave * t + lane;
    if (v < v1) {
      const uint32_t lo = (uint32_t)(16 * v) - mo;               // first byte of the vector, relative to the group
      uint32_t bi = (lo * invS) >> 20;
      if (bi * (uint32_t)S > lo) --bi;
      const uint32_t q = lo - bi * (uint32_t)S;                  // ... relative to its board
      const bool p3 = q <= (uint32_t)(4 * P - 1) && q + 15u >= (uint32_t)(3 * P);
      const bool again = ((redo >> bi) & 1u) || (q + 15u >= (uint32_t)S && ((redo >> (bi + 1u)) & 1u));
      if (LATE ? (p3 || again) : !p3) {
        const uint2 l2 = lut[bb[2 * v]], h2 = lut[bb[2 * v + 1]];
        u4v o;
        o.x = l2.x; o.y = l2.y; o.z = h2.x; o.w = h2.y;
        gv[v] = o;
      }
    }
  }
  if (LATE && r_lo == 0) {   // the ragged vectors at either end of the group (shared with the neighbouring groups): single bytes
    const int head = mo ? 16 - (int)mo : 0, tail = nbits & 15;
    int j = -1;
    if (lane < 16) { if (lane < head) j = lane; }
    else if (lane < 32 && lane - 16 < tail) j = nbrd * S - tail + (lane - 16);
    if (j >= 0 && j < nbrd * S) {
      const uint32_t qq = mo + (uint32_t)j;
      g[j] = (uint8_t)((bs[qq >> 5] >> (qq & 31u)) & 1u);
    }
  }
}

// gogame.batch_next_states (gym_go/gogame.py:90-150; per game next_state :34-87), N == R, out of place, 32 boards per wave.
// EARLY: the output in two instalments (see emit_build), else one write-back at the end.  dbg (A/B builds): bit 0 no
// stores, bit 1 one flood pass instead of eleven, bit 2 no plane loads - to take the launch apart; 0 in the shipped library.
template <int R, bool EARLY>
__global__ __launch_bounds__(kWave, 2) void k_next_states32(const uint8_t *__restrict__ in, const int32_t *__restrict__ actions,
                                                            uint8_t *__restrict__ out, int32_t *__restrict__ status,
                                                            int64_t B, int canonical, int dbg) {
  constexpr int N = R, P = R * R, S = 6 * P, RS = Lds5<R>::RS, RPL = (R + 3) / 4, PL = kNB5 * RS;
  constexpr uint32_t full = (1u << R) - 1u;
  constexpr uint32_t inv16 = (65536u + R - 1u) / R;
  __shared__ __attribute__((aligned(16))) uint32_t lds[Lds5<R>::kTotal];
  uint32_t *rows = lds + Lds5<R>::kRows;
  uint32_t *flagsv = lds + Lds5<R>::kFlags;
  const int lane = threadIdx.x, c = lane & 1, bl = lane >> 1;
  const int64_t ngroups = (B + kNB5 - 1) / kNB5;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t b_first = grp * kNB5;
    const bool on = b_first + bl < B;
    const int64_t b = on ? b_first + bl : B - 1;
    const uint8_t *gi = in + b * (int64_t)S;
    GG_PROF_DECL;
    // ---------------------------------------------------------------- in: own plane, the four flag bytes, the move
    uint32_t m[R];
    const int a = actions[b];
    const bool in_range = a >= 0 && a <= P;
    const bool is_pass = a == P;
    const uint32_t f_turn = gi[2 * P], f_inv = gi[3 * P + ((in_range && !is_pass) ? a : 0)], f_pass = gi[4 * P], f_done = gi[5 * P];
    if (!(dbg & 4)) load_plane_rows<R>(gi + c * P, m);
    else {
#pragma unroll
      for (int r = 0; r < R; ++r) m[r] = 0u;
    }
    const bool illegal = !in_range || (!is_pass && f_inv != 0u);          // gogame.py:59
    const bool moving = !illegal && !is_pass;
    const int pl = (int)(f_turn & 1u);
    const bool mine = c == pl;
    int ar = 0, ac = 0;
    if (moving) split_action(a, N, inv16, ar, ac);
    const uint32_t bit = moving ? (1u << ac) : 0u;
    const uint32_t onehot = moving ? (1u << ar) : 0u;
    // the new stone, on the mover's lane; Q = the plane that holds just that stone (both lanes)
    uint32_t Q[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      Q[r] = (uint32_t)__builtin_amdgcn_sbfe((int)onehot, r, 1) & bit;
      if (mine) m[r] |= Q[r];
    }
    uint32_t e[R];   // empty points of the position with the new stone
#pragma unroll
    for (int r = 0; r < R; ++r) e[r] = full & ~(m[r] | dpp0<QP_SWAP>(m[r]));
    // state_utils.adj_data: `surrounded` = every on-board neighbour of the new stone holds an opponent stone; evaluated
    // on the opponent's lane (its m are the opponent's stones), handed to the partner below
    bool boxed;
    {
      uint32_t dq[R], acc = 0;
      dilate_regs<R>(Q, dq);
#pragma unroll
      for (int r = 0; r < R; ++r) acc |= dq[r] & full & ~m[r];
      boxed = acc == 0u;
    }
    // ---------------------------------------------------------------- out, first instalment: everything but the mask
    const uint32_t passed = is_pass ? 1u : 0u;
    const uint32_t done = (f_done != 0u || (is_pass && f_pass != 0u)) ? 1u : 0u;
    uint32_t nturn = 1u - (uint32_t)pl;
    const bool swap = canonical && nturn == 1u;     // canonical_form (gogame.py:313-321): white to move -> colours swapped
    if (swap) nturn = 0u;
    const int nbrd = (int)((B - b_first) < kNB5 ? (B - b_first) : kNB5);
    uint32_t *const pr = rows + (swap ? 1 - c : c) * PL + bl * RS;
    WAVE_SYNC();
#pragma unroll
    for (int r = 0; r < R; ++r) pr[r] = m[r];
    if (c == 0) flagsv[bl] = nturn | (passed << 1) | (done << 2);
    WAVE_SYNC();
    uint2 *lut = reinterpret_cast<uint2 *>(lds + Lds5<R>::kGrpLut);
    uint32_t *bs0 = lds + Lds5<R>::kGrpBits, *bs1 = bs0 + Lds5<R>::kGrpWords;
    uint8_t *const g0 = out + b_first * (int64_t)S, *const g1 = g0 + 16 * (int64_t)S;
    const int nh0 = nbrd < 16 ? nbrd : 16, nh1 = nbrd - nh0;
    for (int e_ = lane; e_ < 256; e_ += kWave)
      lut[e_] = make_uint2(__umul24((uint32_t)e_ & 15u, 0x204081u) & 0x01010101u, __umul24((uint32_t)e_ >> 4, 0x204081u) & 0x01010101u);
    if (EARLY) {
      emit_build<R, false>(g0, nh0, rows, PL, RS, flagsv, bs0, lane);
      if (nh1 > 0) emit_build<R, false>(g1, nh1, rows + 16 * RS, PL, RS, flagsv + 16, bs1, lane);
    }
    constexpr int kRounds = emit_rounds<R>();                       // store rounds per group
    constexpr int kPerPass = (2 * kRounds + kCwClasses - 1) / kCwClasses;
    GG_PROF(0);   // load + placement + the bit-strings of the first instalment
    // ---------------------------------------------------------------- eleven passes: class k of all 64 units at once
    uint32_t c0[R], c1[R], c2[R], c3[R];
#pragma unroll
    for (int r = 0; r < R; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0u;
    {
      uint32_t mrev[R];
#pragma unroll
      for (int r = 0; r < R; ++r) mrev[r] = __brev(m[r]);
#pragma unroll 1
      for (int k = 0; k < ((dbg & 2) ? 1 : kCwClasses); ++k) {
        uint32_t f[R], g[R];
        {
          uint32_t ee[R + 1];
#pragma unroll
          for (int r = 0; r < R; ++r) ee[r] = e[r] & kCw.m[k][r];   // (wave-uniform class mask: scalar operand)
          ee[R] = 0;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const uint32_t x = r > 0 ? B3(shl1(ee[r]), ee[r] >> 1, ee[r - 1], T_OR3) : (shl1(ee[r]) | (ee[r] >> 1));
            f[r] = B3(m[r], x, ee[r + 1], T_AND_OR2);
          }
        }
        GG_PROF(1);   // seeds
        flood2_serial_regs<R>(m, mrev, f, g);
        GG_PROF(2);   // floods
        // bit-sliced count of the floods that reach each stone (a ripple increment per pass)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint32_t k0 = c0[r] & g[r];
          c0[r] ^= g[r];
          const uint32_t k1 = c1[r] & k0;
          c1[r] ^= k0;
          const uint32_t k2 = c2[r] & k1;
          c2[r] ^= k1;
          c3[r] ^= k2;
        }
        GG_PROF(3);   // counters
        // a slice of the first instalment's stores: rounds [k kPerPass, (k + 1) kPerPass) of the two groups, one after the other
        if (EARLY && !(dbg & 1)) {
          const int t0 = k * kPerPass, t1 = t0 + kPerPass;
          if (t0 < kRounds) emit_store_rounds<R, false>(g0, nh0, bs0, lut, 0u, lane, t0, t1 < kRounds ? t1 : kRounds);
          if (t1 > kRounds && nh1 > 0)
            emit_store_rounds<R, false>(g1, nh1, bs1, lut, 0u, lane, t0 > kRounds ? t0 - kRounds : 0, t1 - kRounds < kRounds ? t1 - kRounds : kRounds);
        }
      }
    }
    uint32_t alive[R], multi[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      alive[r] = B3(c0[r], c1[r], c2[r], T_OR3) | c3[r];
      multi[r] = B3(c3[r], c2[r], c1[r], T_OR_AND);          // >= 6: 8s | (4s & 2s)
    }
    // ---------------------------------------------------------------- captures, ko, class patch
    int ko_r = -1;
    uint32_t ko_bit = 0;
    uint32_t ndead = 0;
    {
      uint32_t dead[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        dead[r] = (!mine && moving) ? (m[r] & ~alive[r]) : 0u;   // opponent groups left without a liberty
        ndead += (uint32_t)__popc(dead[r]);
      }
      if (__ballot(ndead != 0u)) {   // some board of the wave captures (with 32 boards: nearly always)
        // gogame.py:72-75: ko iff exactly one stone died and the new stone is boxed in (both known on the opponent's lane)
        uint32_t krow = 0, kcols = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          krow += dead[r] ? (uint32_t)r : 0u;
          kcols |= dead[r];
          m[r] &= ~dead[r];
        }
        if (ndead == 1u && boxed) { ko_r = (int)krow; ko_bit = kcols; }
        // No second analysis: removing the dead stones only gives liberties to the MOVER's groups next to them (see
        // step_core2 in gg_v2.h): G0, the mover's stones without a liberty (the new stone's group), gets exactly the
        // captured points next to it; a group in atari next to a captured stone now has >= 2.  (Arrays are consumed as
        // soon as they are complete: at most seven row sets are live at any point.)
        uint32_t deadp[R];
#pragma unroll
        for (int r = 0; r < R; ++r) deadp[r] = dpp0<QP_SWAP>(dead[r]);   // on the mover's lane: the captured stones
        {
          uint32_t G0[R], t[R], n0 = 0;
#pragma unroll
          for (int r = 0; r < R; ++r) G0[r] = mine ? (m[r] & ~alive[r]) : 0u;
          dilate_regs<R>(G0, t);
#pragma unroll
          for (int r = 0; r < R; ++r) n0 += (uint32_t)__popc(t[r] & deadp[r]);
          const uint32_t g0m = n0 >= 2u ? ~0u : 0u;
#pragma unroll
          for (int r = 0; r < R; ++r) multi[r] |= G0[r] & g0m;   // (G0 has no liberty: it is not part of the atari set below)
        }
        uint32_t am[R], f[R], anyf = 0;
        {
          uint32_t t[R];
          dilate_regs<R>(deadp, t);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            am[r] = mine ? (m[r] & alive[r] & ~multi[r]) : 0u;
            f[r] = t[r] & am[r];
            anyf |= f[r];
          }
        }
        if (__ballot(anyf != 0u)) {
          uint32_t amrev[R], g[R];
#pragma unroll
          for (int r = 0; r < R; ++r) amrev[r] = __brev(am[r]);
          flood2_serial_regs<R>(am, amrev, f, g);
#pragma unroll
          for (int r = 0; r < R; ++r) multi[r] |= g[r];
        }
      }
    }
    GG_PROF(4);   // captures + patch
    // ---------------------------------------------------------------- the next mover's invalid moves, on ITS lane
    // (state_utils.compute_invalid_moves, point-wise form of gg_v2.h's invalid_from2): nx = this lane's stones (the
    // opponent of the mover moves next), pl = the partner's
    uint32_t inv[R];
    {
      uint32_t x[R], e2[R], nb[R];
      const uint32_t kohot = ko_r >= 0 ? (1u << ko_r) : 0u;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t pm = dpp0<QP_SWAP>(m[r]), pmulti = dpp0<QP_SWAP>(multi[r]);
        e2[r] = full & ~(m[r] | pm);
        x[r] = B3(e2[r], m[r] & multi[r], pm & ~pmulti, T_OR3);
      }
      dilate_regs<R>(x, nb);
#pragma unroll
      for (int r = 0; r < R; ++r)
        inv[r] = (full & ~(e2[r] & nb[r])) | ((uint32_t)__builtin_amdgcn_sbfe((int)kohot, r, 1) & ko_bit);
    }
    // ---------------------------------------------------------------- out, second instalment: the mask + changed stones
    const uint64_t capt = __ballot(ndead != 0u);     // (set on the opponent's lane of a board that lost stones)
    uint64_t cx = (capt | (capt >> 1)) & 0x5555555555555555ull;   // bit 2 i: board i of the wave captured -> bit i
    cx = (cx | (cx >> 1)) & 0x3333333333333333ull;
    cx = (cx | (cx >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    cx = (cx | (cx >> 4)) & 0x00FF00FF00FF00FFull;
    cx = (cx | (cx >> 8)) & 0x0000FFFF0000FFFFull;
    const uint32_t redo32 = (uint32_t)(cx | (cx >> 16));
    WAVE_SYNC();
    if (ndead != 0u) {
#pragma unroll
      for (int r = 0; r < R; ++r) pr[r] = m[r];       // the stones that are left
    }
    if (!mine) {   // the next mover's lane owns the mask
      uint32_t *pi = rows + 2 * PL + bl * RS;
#pragma unroll
      for (int r = 0; r < R; ++r) pi[r] = inv[r];
    }
    WAVE_SYNC();
    const uint32_t redo = EARLY ? redo32 : ~0u;      // one instalment: every vector is written now
    emit_build<R, true>(g0, nh0, rows, PL, RS, flagsv, bs0, lane);
    if (nh1 > 0) emit_build<R, true>(g1, nh1, rows + 16 * RS, PL, RS, flagsv + 16, bs1, lane);
    if (!(dbg & 1)) {
      emit_store_rounds<R, true>(g0, nh0, bs0, lut, redo & 0xFFFFu, lane, 0, kRounds);
      if (nh1 > 0) emit_store_rounds<R, true>(g1, nh1, bs1, lut, redo >> 16, lane, 0, kRounds);
    }
    // a refused move: the input row passes through (stores of one wave to one address land in program order)
    if (__ballot(on && illegal)) {
#pragma unroll 1
      for (int i = 0; i < kNB5; ++i) {
        const int ill = __shfl((int)(on && illegal), 2 * i);
        if (ill) {
          const uint8_t *src = in + (b_first + i) * (int64_t)S;
          uint8_t *dst = out + (b_first + i) * (int64_t)S;
          for (int j = lane; j < S; j += kWave) dst[j] = src[j];
        }
      }
    }
    if (status && on && c == 0) status[b] = illegal ? GG_STATUS_ILLEGAL : GG_STATUS_OK;
    WAVE_SYNC();
    GG_PROF(6);   // emit
    GG_PROF_FLUSH;
  }
}

}  // namespace gg
