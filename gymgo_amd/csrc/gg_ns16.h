// gg_ns16.h - the from-scratch step kernels for big batches of full-size boards (k_next_states16: gogame.batch_next_states;
// k_env_step16: the batched GoEnv.step and the one-ply rollout on byte planes; k_invalid_mask16): SIXTEEN BOARDS PER
// WAVEFRONT, one lane per (board, colour, class half), every liberty class from scratch with (nearly) all 64 lanes flooding.
//
// The two-boards-per-wave kernel (k_next_states2, gg_v2.h) runs the 22 floods of a board (11 liberty classes x 2 colours,
// constant-weight code) side by side: 44 of 64 lanes carry a flood, and every point-wise rule runs in the "row per lane"
// layout where 19 of 32 lanes do work.  Here the floods run CLASS-MAJOR: lane (b, h, c) = 4 b + 2 h + c holds all rows of
// colour c of board b in registers for the whole step, pass j floods class 2 j + h of its unit - six passes for sixteen
// boards, 0.375 wave-floods per board instead of 0.5, 59 of 64 lanes busy on average - and counts, in three bit-sliced
// counter planes of its own, how many of its floods reach each of its stones; the two halves of a unit add their counts
// once at the end (>= 1: the group has a liberty, >= 6: it has two or more; gg_v2.h explains the code).  Nothing goes
// through LDS between the passes, and every other step of gogame.next_state - placing the stone,
// state_utils.update_pieces (gym_go/state_utils.py:159-180), the ko rule (gym_go/gogame.py:72-75),
// state_utils.compute_invalid_moves (:24-83) - runs on the same 19-rows-per-lane registers: vertical neighbours are
// neighbouring registers, the other colour of the board sits in the partner lane (one DPP quad_perm swap per row); the two
// class halves of a unit carry the same rows and do the point-wise work twice, in lock-step, at no extra instruction.
// Boards come in by one aligned 16-byte vector stream per lane (its own plane, 361 bytes), bit-packed with v_dot4; they
// leave as one contiguous byte range per group through a bit-string in LDS and aligned 1 KB blocks.
//
// This is the 16-board, three-waves-per-SIMD form of the 32-board kernel measured in round 3 (tools/exp/attic/ (round 3, in git history): 45 % fewer
// instructions than the two-board kernel, but 256 VGPRs = two lock-step wave-iterations per SIMD at 65 536 boards, whose
// loads and stores nothing overlapped).  Three waves per SIMD and the arbiter's oldest-wave-first order change that: the
// groups of a SIMD are split 2 : 1 : 1 by wave age (pair_span, gg_common.h), so the oldest wave is storing its first
// group while the younger ones still flood theirs.  (13x13 and 9x9 need fewer registers: four waves per SIMD, one
// workgroup per group; so does 19x19 on a device where the kernel does not get its three waves.)  Dispatch: gg_kernels.hip,
// `use_ns16` - batches of exactly 9x9 / 13x13 / 19x19 boards from a number of groups per SIMD on that was measured per
// entry point and board size (gg_batch_next_states: 65 536 boards of 19x19, 32 768 of 13x13 / 9x9; gg_batch_env_step and the
// one-ply rollout: 49 152 / 32 768 / 16 384; gg_batch_invalid_mask and gg_batch_track_states (k_invalid_mask16, k_track16): 19x19
// from 65 536 - four groups per SIMD, split 2 : 1 : 1 by wave age - 13x13 from 32 768, 9x9 from 16 384).
#pragma once
#include "gg_v4.h"

namespace gg {

constexpr int kNB16 = 16;
#ifdef GG_AB
__device__ int gg_ns16_dbg;   // A/B builds: bit 0 no stores, bit 1 one flood pass instead of six, bit 2 no plane loads
#define GG_NS16_DBG(bit) ((dbg_ >> (bit)) & 1)
#else
#define GG_NS16_DBG(bit) 0
#endif

// waves per SIMD the kernel is compiled for: 19x19 needs 164 registers (three waves), the smaller boards fit four
template <int R> struct Ns16Waves { static constexpr int value = R > 13 ? 3 : 4; };

template <int R>
struct Lds16 {
  static constexpr int RS = Cfg<R>::kRowStride;
  static_assert(RS > R, "word R of a board's black rows holds its flag word");
  static constexpr int kRows = 0;                                   // [3][16][RS]: black, white, invalid rows of the results (word R of a black row block: bit 0 turn, 1 passed, 2 done)
  static constexpr int kCwt = kRows + 3 * kNB16 * RS;               // [12][20]: the class masks (11 classes + zeros)
  static constexpr int kGrpBits = kCwt + (kCwClasses + 1) * 20;     // the bit-string of the group ...
  static constexpr int kGrpWords = ((15 + kNB16 * 6 * R * R + 31) / 32 + 4) & ~3;
  static constexpr int kGrpLut = kGrpBits + kGrpWords;              // ... and the 8 bits -> 8 bytes table
  static constexpr int kTotal = kGrpLut + 512;
  static_assert(kTotal * 4 * 4 * Ns16Waves<R>::value <= 160 * 1024, "LDS of the resident workgroups of a CU");
};

constexpr int QP_COLOUR = 0xB1;   // quad_perm [1,0,3,2]: the other colour of the board
constexpr int QP_HALF = 0x4E;     // quad_perm [2,3,0,1]: the other class half of the unit

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
    if (it > 0 || R <= 9) {   // (9x9: the second sweep usually closes the fill, see flood2_serial)
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

// The group's boards are one contiguous byte range [g, g + nbrd * 6 N^2): it is built as a bit-string in LDS (bit i =
// byte i of the range, counted from the 16-byte boundary below its first byte) from the row masks
// rows[p * PL + board * RS + r] (p = 0 black, 1 white, 2 invalid; word R of a black block = the flags), and leaves as aligned 16-byte
// vectors, 64 lanes x 16 B = 1 KB per instruction, through the 8 bits -> 8 bytes table (emit_group of gg_v4.h with the
// mask rows read from LDS instead of the quad layout's registers).
// NP = 6: whole states; NP = 1: one plane per board, its rows in rows[0 * PL ...] (the invalid-move masks of a group)
template <int R, int NP = 6>
__device__ __forceinline__ void emit_rows16(uint8_t *g, int nbrd, const uint32_t *rows, int PL, int RS,
                                            uint32_t *bs, const uint2 *lut, int lane, bool nostore = false) {
  constexpr int N = R, P = R * R, S = NP * P, RPL = (R + 3) / 4;
  const uint32_t mo = (uint32_t)((uintptr_t)g & 15u);
  const int nbits = (int)mo + nbrd * S;
  const int q4 = lane >> 2, r04 = RPL * (lane & 3);
  for (int i = lane; i < (nbits + 31) / 32 + 1; i += kWave) bs[i] = 0;
  WAVE_SYNC();
  if (q4 < nbrd) {
    const uint32_t fl = NP == 6 ? rows[q4 * RS + R] : 0u;
    constexpr uint32_t fullrow = (1u << N) - 1u;
    const uint32_t tp = (fl & 1u) ? fullrow : 0u, pp = (fl & 2u) ? fullrow : 0u, dp = (fl & 4u) ? fullrow : 0u;
    const uint32_t base = mo + (uint32_t)(q4 * S);
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const int rr = r04 + r;
      if (rr < N) {
        const uint32_t rw[6] = {rows[0 * PL + q4 * RS + rr], NP == 6 ? rows[1 * PL + q4 * RS + rr] : 0u, tp,
                                NP == 6 ? rows[2 * PL + q4 * RS + rr] : 0u, pp, dp};
#pragma unroll
        for (int p = 0; p < NP; ++p) {
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
  uint8_t *ga = g - mo;
  const int v0 = mo ? 1 : 0, v1 = nbits >> 4;
  const uint8_t *bb = reinterpret_cast<const uint8_t *>(bs);
  typedef uint32_t u4v __attribute__((ext_vector_type(4)));
  u4v *gv = reinterpret_cast<u4v *>(__builtin_assume_aligned(ga, 16));
  for (int v = v0 + lane; v < (nostore ? v0 : v1); v += kWave) {
    const uint2 l2 = lut[bb[2 * v]], h2 = lut[bb[2 * v + 1]];
    u4v o;
    o.x = l2.x; o.y = l2.y; o.z = h2.x; o.w = h2.y;
    gv[v] = o;
  }
  // the ragged vectors at either end of the group (shared with the neighbouring groups): single bytes
  const int head = mo ? 16 - (int)mo : 0, tail = nbits & 15;
  int j = -1;
  if (lane < 16) { if (lane < head) j = lane; }
  else if (lane < 32 && lane - 16 < tail) j = nbrd * S - tail + (lane - 16);
  if (j >= 0 && j < nbrd * S) {
    const uint32_t qq = mo + (uint32_t)j;
    g[j] = (uint8_t)((bs[qq >> 5] >> (qq & 31u)) & 1u);
  }
}

// gogame.batch_next_states (gym_go/gogame.py:90-150; per game next_state :34-87), N == R, out of place, 16 boards per wave
template <int R>
__global__ __launch_bounds__(kWave, Ns16Waves<R>::value) void k_next_states16(const uint8_t *__restrict__ in, const int32_t *__restrict__ actions,
                                                            uint8_t *__restrict__ out, int32_t *__restrict__ status,
                                                            int64_t B, int canonical, AgeSplit age) {
  constexpr int N = R, P = R * R, S = 6 * P, RS = Lds16<R>::RS, PL = kNB16 * RS;
  constexpr uint32_t full = (1u << R) - 1u;
  constexpr uint32_t inv16 = (65536u + R - 1u) / R;
  __shared__ __attribute__((aligned(16))) uint32_t lds[Lds16<R>::kTotal];
  uint32_t *rows = lds + Lds16<R>::kRows;
  uint32_t *cwt = lds + Lds16<R>::kCwt;
  uint2 *lut = reinterpret_cast<uint2 *>(lds + Lds16<R>::kGrpLut);
  {
    const int l0 = threadIdx.x;
    CwRegs cw_;   // (all reads of the class rows in flight at once, the spread table built meanwhile: cw_table_issue, gg_v2.h)
    cw_table_issue(cw_, l0);
    for (int e_ = l0; e_ < 256; e_ += kWave)
      lut[e_] = make_uint2(__umul24((uint32_t)e_ & 15u, 0x204081u) & 0x01010101u, __umul24((uint32_t)e_ >> 4, 0x204081u) & 0x01010101u);
    cw_rows_commit(cwt, cw_, l0);
  }
  WAVE_SYNC();
#ifdef GG_AB
  const int dbg_ = gg_ns16_dbg;
#endif
  const int64_t ngroups = (B + kNB16 - 1) / kNB16;
  const PairSpan span = pair_span(ngroups, age);
  for (int64_t grp = span.first; grp < span.end; grp += span.stride) {
    // (lane-derived values from a fresh lane id inside the loop: hoisted, they are spilled across the passes)
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    const int c = lane & 1, h = (lane >> 1) & 1, bl = lane >> 2;
    const int64_t b_first = grp * kNB16;
    const bool on = b_first + bl < B;
    const int64_t b = on ? b_first + bl : B - 1;
    const uint8_t *gi = in + b * (int64_t)S;
    // ---------------------------------------------------------------- in: own plane, the four flag bytes, the move
    uint32_t m[R];
    const int a = actions[b];
    const bool in_range = a >= 0 && a <= P;
    const bool is_pass = a == P;
    const uint32_t f_turn = gi[2 * P], f_inv = gi[3 * P + ((in_range && !is_pass) ? a : 0)], f_pass = gi[4 * P], f_done = gi[5 * P];
    if (!GG_NS16_DBG(2)) load_plane_rows<R>(gi + c * P, m);
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
    // the new stone, on the mover's lanes; `boxed` (state_utils.adj_data's `surrounded`: every on-board neighbour of the
    // new stone holds an opponent stone) is evaluated on the opponent's lanes, whose m are the opponent's stones
    bool boxed;
    {
      uint32_t Q[R], dq[R], acc = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        Q[r] = (uint32_t)__builtin_amdgcn_sbfe((int)onehot, r, 1) & bit;
        if (mine) m[r] |= Q[r];
      }
      dilate_regs<R>(Q, dq);
#pragma unroll
      for (int r = 0; r < R; ++r) acc |= dq[r] & full & ~m[r];
      // (pinned here: left to itself the compiler sinks this block below the passes, where `boxed` is used, and carries
      // nineteen half-computed rows across them in scratch)
      asm volatile("" : "+v"(acc));
      boxed = acc == 0u;
    }
    const uint32_t passed = is_pass ? 1u : 0u;
    const uint32_t done = (f_done != 0u || (is_pass && f_pass != 0u)) ? 1u : 0u;
    uint32_t nturn = 1u - (uint32_t)pl;
    const bool swap = canonical && nturn == 1u;     // canonical_form (gogame.py:313-321): white to move -> colours swapped
    if (swap) nturn = 0u;
    const int nbrd = (int)((B - b_first) < kNB16 ? (B - b_first) : kNB16);
    // ---------------------------------------------------------------- six passes: class 2 j + h of all 32 units at once
    uint32_t alive[R], multi[R];
    {
      uint32_t c0[R], c1[R], c2[R], mrev[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        c0[r] = c1[r] = c2[r] = 0u;
        mrev[r] = __brev(m[r]);
      }
#pragma unroll 1
      for (int j = 0; j < (GG_NS16_DBG(1) ? 1 : (kCwClasses + 1) / 2); ++j) {
        uint32_t f[R], g[R];
        {
          // empty points of the position with the new stone, of this lane's class (2 j + h = 11: no such class, zeros)
          uint32_t ee[R + 1];
          const uint4 *pc = reinterpret_cast<const uint4 *>(cwt + (2 * j + h) * 20);
#pragma unroll
          for (int i = 0; i < (R + 3) / 4; ++i) {
            const uint4 d = pc[i];
            if (4 * i < R) ee[4 * i] = d.x;
            if (4 * i + 1 < R) ee[4 * i + 1] = d.y;
            if (4 * i + 2 < R) ee[4 * i + 2] = d.z;
            if (4 * i + 3 < R) ee[4 * i + 3] = d.w;
          }
#pragma unroll
          for (int r = 0; r < R; ++r) ee[r] = B3(ee[r], m[r], dpp0<QP_COLOUR>(m[r]), TA & ~(TB | TC) & 0xFF) & full;
          ee[R] = 0;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const uint32_t x = r > 0 ? B3(shl1(ee[r]), ee[r] >> 1, ee[r - 1], T_OR3) : (shl1(ee[r]) | (ee[r] >> 1));
            f[r] = B3(m[r], x, ee[r + 1], T_AND_OR2);
          }
        }
        flood2_serial_regs<R>(m, mrev, f, g);
        // bit-sliced count of the floods that reach each stone (a ripple increment per pass; <= 6 floods per lane)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint32_t k0 = c0[r] & g[r];
          c0[r] ^= g[r];
          const uint32_t k1 = c1[r] & k0;
          c1[r] ^= k0;
          c2[r] |= k1;
        }
      }
      // the unit's count = this half's + the other half's (<= 11): alive = count >= 1, multi = count >= 6
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t c2r = c2[r];
        const uint32_t p0 = dpp0<QP_HALF>(c0[r]), p1 = dpp0<QP_HALF>(c1[r]), p2 = dpp0<QP_HALF>(c2r);
        const uint32_t k0 = c0[r] & p0;
        const uint32_t s1 = B3(c1[r], p1, k0, TA ^ TB ^ TC);
        const uint32_t k1 = B3(c1[r], p1, k0, (TA & TB) | (TA & TC) | (TB & TC));
        const uint32_t s2 = B3(c2r, p2, k1, TA ^ TB ^ TC);
        const uint32_t s3 = B3(c2r, p2, k1, (TA & TB) | (TA & TC) | (TB & TC));
        alive[r] = B3(c0[r], c1[r], c2r, T_OR3) | B3(p0, p1, p2, T_OR3);
        multi[r] = B3(s3, s2, s1, T_OR_AND);          // >= 6: 8s | (4s & 2s)
      }
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
      if (__ballot(ndead != 0u)) {   // some board of the wave captures
        // gogame.py:72-75: ko iff exactly one stone died and the new stone is boxed in (both known on the opponent's lanes)
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
        // captured points next to it; a group in atari next to a captured stone now has >= 2.
        uint32_t deadp[R];
#pragma unroll
        for (int r = 0; r < R; ++r) deadp[r] = dpp0<QP_COLOUR>(dead[r]);   // on the mover's lanes: the captured stones
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
    // ---------------------------------------------------------------- the next mover's invalid moves, on ITS lanes
    // (state_utils.compute_invalid_moves, point-wise form of gg_v2.h's invalid_from2): nx = this lane's stones (the
    // opponent of the mover moves next), pl = the partner's
    uint32_t inv[R];
    {
      uint32_t x[R], e2[R], nb[R];
      const uint32_t kohot = ko_r >= 0 ? (1u << ko_r) : 0u;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t pm = dpp0<QP_COLOUR>(m[r]), pmulti = dpp0<QP_COLOUR>(multi[r]);
        e2[r] = full & ~(m[r] | pm);
        x[r] = B3(e2[r], m[r] & multi[r], pm & ~pmulti, T_OR3);
      }
      dilate_regs<R>(x, nb);
#pragma unroll
      for (int r = 0; r < R; ++r)
        inv[r] = (full & ~(e2[r] & nb[r])) | ((uint32_t)__builtin_amdgcn_sbfe((int)kohot, r, 1) & ko_bit);
    }
    // ---------------------------------------------------------------- out: the group as one contiguous byte range
    WAVE_SYNC();
    if (h == 0) {
      uint32_t *const pr = rows + (swap ? 1 - c : c) * PL + bl * RS;
#pragma unroll
      for (int r = 0; r < R; ++r) pr[r] = m[r];
      if (!mine) {   // the next mover's lane owns the mask
        uint32_t *pi = rows + 2 * PL + bl * RS;
#pragma unroll
        for (int r = 0; r < R; ++r) pi[r] = inv[r];
      }
      if (c == 0) rows[bl * RS + R] = nturn | (passed << 1) | (done << 2);
    }
    WAVE_SYNC();
    emit_rows16<R>(out + b_first * (int64_t)S, nbrd, rows, PL, RS, lds + Lds16<R>::kGrpBits, lut, lane, GG_NS16_DBG(0));
    // a refused move: the input row passes through (stores of one wave to one address land in program order)
    if (__ballot(on && illegal)) {
#pragma unroll 1
      for (int i = 0; i < kNB16; ++i) {
        const int ill = __shfl((int)(on && illegal), 4 * i);
        if (ill) {
          const uint8_t *src = in + (b_first + i) * (int64_t)S;
          uint8_t *dst = out + (b_first + i) * (int64_t)S;
          for (int jj = lane; jj < S; jj += kWave) dst[jj] = src[jj];
        }
      }
    }
    if (status && on && (lane & 3) == 0) status[b] = illegal ? GG_STATUS_ILLEGAL : GG_STATUS_OK;
    WAVE_SYNC();
  }
}

// state_utils.batch_compute_invalid_moves (gym_go/state_utils.py:86-156; per game :24-83) for big batches, the same
// class-major analysis without a move: the lanes of the colour to move (plane 2) derive the mask, `ko` (int32 [B] or
// nullptr: the point a ko forbids, -1 = none) is added as in gg_batch_invalid_mask's two-board kernel.
template <int R>
__global__ __launch_bounds__(kWave, Ns16Waves<R>::value) void k_invalid_mask16(const uint8_t *__restrict__ states,
                                                                               const int32_t *__restrict__ ko,
                                                                               uint8_t *__restrict__ mask, int64_t B, AgeSplit age) {
  constexpr int N = R, P = R * R, S = 6 * P, RS = Lds16<R>::RS, PL = kNB16 * RS;
  constexpr uint32_t full = (1u << R) - 1u;
  constexpr uint32_t inv16 = (65536u + R - 1u) / R;
  __shared__ __attribute__((aligned(16))) uint32_t lds[Lds16<R>::kTotal];
  uint32_t *rows = lds + Lds16<R>::kRows;
  uint32_t *cwt = lds + Lds16<R>::kCwt;
  uint2 *lut = reinterpret_cast<uint2 *>(lds + Lds16<R>::kGrpLut);
  {
    const int l0 = threadIdx.x;
    CwRegs cw_;   // (all reads of the class rows in flight at once, the spread table built meanwhile: cw_table_issue, gg_v2.h)
    cw_table_issue(cw_, l0);
    for (int e_ = l0; e_ < 256; e_ += kWave)
      lut[e_] = make_uint2(__umul24((uint32_t)e_ & 15u, 0x204081u) & 0x01010101u, __umul24((uint32_t)e_ >> 4, 0x204081u) & 0x01010101u);
    cw_rows_commit(cwt, cw_, l0);
  }
  WAVE_SYNC();
  const int64_t ngroups = (B + kNB16 - 1) / kNB16;
  const PairSpan span = pair_span(ngroups, age);
  for (int64_t grp = span.first; grp < span.end; grp += span.stride) {
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    const int c = lane & 1, h = (lane >> 1) & 1, bl = lane >> 2;
    const int64_t b_first = grp * kNB16;
    const bool on = b_first + bl < B;
    const int64_t b = on ? b_first + bl : B - 1;
    const uint8_t *gi = states + b * (int64_t)S;
    uint32_t m[R];
    load_plane_rows<R>(gi + c * P, m);
    uint32_t multi[R];
    {
      uint32_t c0[R], c1[R], c2[R], mrev[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        c0[r] = c1[r] = c2[r] = 0u;
        mrev[r] = __brev(m[r]);
      }
#pragma unroll 1
      for (int j = 0; j < (kCwClasses + 1) / 2; ++j) {
        uint32_t f[R], g[R];
        {
          uint32_t ee[R + 1];
          const uint4 *pc = reinterpret_cast<const uint4 *>(cwt + (2 * j + h) * 20);
#pragma unroll
          for (int i = 0; i < (R + 3) / 4; ++i) {
            const uint4 d = pc[i];
            if (4 * i < R) ee[4 * i] = d.x;
            if (4 * i + 1 < R) ee[4 * i + 1] = d.y;
            if (4 * i + 2 < R) ee[4 * i + 2] = d.z;
            if (4 * i + 3 < R) ee[4 * i + 3] = d.w;
          }
#pragma unroll
          for (int r = 0; r < R; ++r) ee[r] = B3(ee[r], m[r], dpp0<QP_COLOUR>(m[r]), TA & ~(TB | TC) & 0xFF) & full;
          ee[R] = 0;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const uint32_t x = r > 0 ? B3(shl1(ee[r]), ee[r] >> 1, ee[r - 1], T_OR3) : (shl1(ee[r]) | (ee[r] >> 1));
            f[r] = B3(m[r], x, ee[r + 1], T_AND_OR2);
          }
        }
        flood2_serial_regs<R>(m, mrev, f, g);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint32_t k0 = c0[r] & g[r];
          c0[r] ^= g[r];
          const uint32_t k1 = c1[r] & k0;
          c1[r] ^= k0;
          c2[r] |= k1;
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t p0 = dpp0<QP_HALF>(c0[r]), p1 = dpp0<QP_HALF>(c1[r]), p2 = dpp0<QP_HALF>(c2[r]);
        const uint32_t k0 = c0[r] & p0;
        const uint32_t s1 = B3(c1[r], p1, k0, TA ^ TB ^ TC);
        const uint32_t k1 = B3(c1[r], p1, k0, (TA & TB) | (TA & TC) | (TB & TC));
        const uint32_t s2 = B3(c2[r], p2, k1, TA ^ TB ^ TC);
        const uint32_t s3 = B3(c2[r], p2, k1, (TA & TB) | (TA & TC) | (TB & TC));
        multi[r] = B3(s3, s2, s1, T_OR_AND);          // >= 6 floods: two or more liberties
      }
    }
    // the mask, on the lanes of the colour to move (invalid_from2 of gg_v2.h, point-wise): a point is playable iff it is
    // empty and next to an empty point, to one of the mover's groups with >= 2 liberties or to an opponent group in atari
    // (the turn byte and the ko point are fetched here, after the passes: held across them they cost scratch)
    WAVE_SYNC();
    const int nbrd = (int)((B - b_first) < kNB16 ? (B - b_first) : kNB16);
    {
      const uint32_t f_turn = gi[2 * P];
      const int k = ko ? ko[b] : -1;
      const bool owner = (uint32_t)c == (f_turn & 1u);
      int kr = -1, kc = 0;
      if (k >= 0 && k < P) split_action(k, N, inv16, kr, kc);
      const uint32_t kohot = kr >= 0 ? (1u << kr) : 0u, kbit = 1u << kc;
      uint32_t x[R], e2[R], nb[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t pm = dpp0<QP_COLOUR>(m[r]), pmulti = dpp0<QP_COLOUR>(multi[r]);
        e2[r] = full & ~(m[r] | pm);
        x[r] = B3(e2[r], m[r] & multi[r], pm & ~pmulti, T_OR3);
      }
      dilate_regs<R>(x, nb);
      if (h == 0 && owner) {
        uint32_t *pi = rows + bl * RS;
#pragma unroll
        for (int r = 0; r < R; ++r)
          pi[r] = (full & ~(e2[r] & nb[r])) | ((uint32_t)__builtin_amdgcn_sbfe((int)kohot, r, 1) & kbit);
      }
    }
    WAVE_SYNC();
    emit_rows16<R, 1>(mask + b_first * (int64_t)P, nbrd, rows, PL, RS, lds + Lds16<R>::kGrpBits, lut, lane);
    WAVE_SYNC();
  }
}

// byte planes -> tracked boards (gg_batch_track_states; k_track of gg_v4.h for big batches of full-size boards): the same
// class-major analysis; lane (h = 0, c) writes its colour's stone rows and their ">= 2 liberties" rows, lane (1, 0) plane 3
// as it stands (it carries the ko point of the last move: not recomputed), lane (1, 1) the flag word.
template <int R>
__global__ __launch_bounds__(kWave, Ns16Waves<R>::value) void k_track16(const uint8_t *__restrict__ states,
                                                                        uint32_t *__restrict__ tracked, int64_t B, AgeSplit age) {
  constexpr int N = R, P = R * R, S = 6 * P, W = 5 * N + 1;
  constexpr uint32_t full = (1u << R) - 1u;
  __shared__ __attribute__((aligned(16))) uint32_t cwt[(kCwClasses + 1) * 20];
  {
    const int l0 = threadIdx.x;
    CwRegs cw_;
    cw_table_issue(cw_, l0);
    cw_rows_commit(cwt, cw_, l0);
  }
  WAVE_SYNC();
  const int64_t ngroups = (B + kNB16 - 1) / kNB16;
  const PairSpan span = pair_span(ngroups, age);
  for (int64_t grp = span.first; grp < span.end; grp += span.stride) {
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    const int c = lane & 1, h = (lane >> 1) & 1, bl = lane >> 2;
    const int64_t b_first = grp * kNB16;
    const bool on = b_first + bl < B;
    const int64_t b = on ? b_first + bl : B - 1;
    const uint8_t *gi = states + b * (int64_t)S;
    uint32_t m[R];
    load_plane_rows<R>(gi + c * P, m);
    uint32_t multi[R];
    {
      uint32_t c0[R], c1[R], c2[R], mrev[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        c0[r] = c1[r] = c2[r] = 0u;
        mrev[r] = __brev(m[r]);
      }
#pragma unroll 1
      for (int j = 0; j < (kCwClasses + 1) / 2; ++j) {
        uint32_t f[R], g[R];
        {
          uint32_t ee[R + 1];
          const uint4 *pc = reinterpret_cast<const uint4 *>(cwt + (2 * j + h) * 20);
#pragma unroll
          for (int i = 0; i < (R + 3) / 4; ++i) {
            const uint4 d = pc[i];
            if (4 * i < R) ee[4 * i] = d.x;
            if (4 * i + 1 < R) ee[4 * i + 1] = d.y;
            if (4 * i + 2 < R) ee[4 * i + 2] = d.z;
            if (4 * i + 3 < R) ee[4 * i + 3] = d.w;
          }
#pragma unroll
          for (int r = 0; r < R; ++r) ee[r] = B3(ee[r], m[r], dpp0<QP_COLOUR>(m[r]), TA & ~(TB | TC) & 0xFF) & full;
          ee[R] = 0;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const uint32_t x = r > 0 ? B3(shl1(ee[r]), ee[r] >> 1, ee[r - 1], T_OR3) : (shl1(ee[r]) | (ee[r] >> 1));
            f[r] = B3(m[r], x, ee[r + 1], T_AND_OR2);
          }
        }
        flood2_serial_regs<R>(m, mrev, f, g);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint32_t k0 = c0[r] & g[r];
          c0[r] ^= g[r];
          const uint32_t k1 = c1[r] & k0;
          c1[r] ^= k0;
          c2[r] |= k1;
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t p0 = dpp0<QP_HALF>(c0[r]), p1 = dpp0<QP_HALF>(c1[r]), p2 = dpp0<QP_HALF>(c2[r]);
        const uint32_t k0 = c0[r] & p0;
        const uint32_t s1 = B3(c1[r], p1, k0, TA ^ TB ^ TC);
        const uint32_t k1 = B3(c1[r], p1, k0, (TA & TB) | (TA & TC) | (TB & TC));
        const uint32_t s2 = B3(c2[r], p2, k1, TA ^ TB ^ TC);
        const uint32_t s3 = B3(c2[r], p2, k1, (TA & TB) | (TA & TC) | (TB & TC));
        multi[r] = B3(s3, s2, s1, T_OR_AND);          // >= 6 floods: two or more liberties
      }
    }
    uint32_t *gp = tracked + b * (int64_t)W;
    if (h == 0) {
      if (on) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          gp[c * N + r] = m[r];
          gp[(3 + c) * N + r] = m[r] & multi[r];
        }
      }
    } else if (c == 0) {
      uint32_t iv[R];
      load_plane_rows<R>(gi + 3 * P, iv);
      if (on) {
#pragma unroll
        for (int r = 0; r < R; ++r) gp[2 * N + r] = iv[r];
      }
    } else if (on) {
      gp[5 * N] = (gi[2 * P] & 1u) | ((gi[4 * P] & 1u) << 1) | ((gi[5 * P] & 1u) << 2);
    }
  }
}

// One GoEnv.step for every game of a batched env on byte planes, in place (gym_go/envs/go_env.py:49-76; the contract of
// k_env_step2, gg_v2.h: auto-reset or refusal of finished games, the action given or drawn uniformly among the valid
// moves, legality, next_state, game_ended, GoEnv.reward :128-149), for big batches: the class-major analysis of
// k_next_states16.  The spare class-half lanes (h = 1) load the invalid-move plane instead of a second copy of their
// colour's stones: they draw the move (valid points, prefix counts and the k-th set bit, all in one lane) or test the
// given one, hand the verdict to their unit, and only then take the stone rows over from the h = 0 lanes.  The
// Tromp-Taylor areas (gym_go/gogame.py:275-300) are one more flood pass of the empty points, seeded next to the lane's
// colour - every step for the heuristic reward (HEUR), only in a wave where a game has just ended for the real one.
// last_actions / steps_done: the outputs of a one-ply gg_batch_rollout, which is this step without the GoEnv outputs.
template <int R, bool HEUR>
__global__ __launch_bounds__(kWave, Ns16Waves<R>::value) void k_env_step16(uint8_t *__restrict__ states, const int32_t *__restrict__ actions,
                                                                           uint64_t *__restrict__ rng, float *__restrict__ rewards,
                                                                           uint8_t *__restrict__ dones, int32_t *__restrict__ status,
                                                                           int32_t *__restrict__ taken, int64_t B, float komi,
                                                                           int auto_reset, AgeSplit age,
                                                                           int32_t *__restrict__ last_actions = nullptr,
                                                                           int64_t *__restrict__ steps_done = nullptr) {
  constexpr int N = R, P = R * R, S = 6 * P, RS = Lds16<R>::RS, PL = kNB16 * RS;
  constexpr uint32_t full = (1u << R) - 1u;
  constexpr uint32_t inv16 = (65536u + R - 1u) / R;
  __shared__ __attribute__((aligned(16))) uint32_t lds[Lds16<R>::kTotal];
  uint32_t *rows = lds + Lds16<R>::kRows;
  uint32_t *cwt = lds + Lds16<R>::kCwt;
  uint2 *lut = reinterpret_cast<uint2 *>(lds + Lds16<R>::kGrpLut);
  {
    const int l0 = threadIdx.x;
    CwRegs cw_;   // (all reads of the class rows in flight at once, the spread table built meanwhile: cw_table_issue, gg_v2.h)
    cw_table_issue(cw_, l0);
    for (int e_ = l0; e_ < 256; e_ += kWave)
      lut[e_] = make_uint2(__umul24((uint32_t)e_ & 15u, 0x204081u) & 0x01010101u, __umul24((uint32_t)e_ >> 4, 0x204081u) & 0x01010101u);
    cw_rows_commit(cwt, cw_, l0);
  }
  WAVE_SYNC();
  const int64_t ngroups = (B + kNB16 - 1) / kNB16;
  const PairSpan span = pair_span(ngroups, age);
  for (int64_t grp = span.first; grp < span.end; grp += span.stride) {
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    const int c = lane & 1, h = (lane >> 1) & 1, bl = lane >> 2;
    const uint32_t hm = h ? ~0u : 0u;
    const int64_t b_first = grp * kNB16;
    const bool on = b_first + bl < B;
    const int64_t b = on ? b_first + bl : B - 1;
    uint8_t *gi = states + b * (int64_t)S;
    // ---------------------------------------------------------------- in: flags, the plane of this lane, the move
    const uint32_t f_turn = gi[2 * P], f_pass = gi[4 * P], f_done = gi[5 * P];
    int a = actions ? actions[b] : 0;
    uint64_t x = actions ? 0ull : rng[b];
    uint32_t m[R];
    load_plane_rows<R>(gi + (h ? 3 : c) * P, m);          // h = 1: the invalid-move plane (plane 3)
    const bool frozen = f_done != 0u && !auto_reset;      // go_env.py:53 "assert not self.done"
    int pl = (int)(f_turn & 1u);
    uint32_t passed = f_pass != 0u ? 1u : 0u, done = f_done != 0u ? 1u : 0u;
    if (done && auto_reset) {                             // GoEnv.reset (:40-47) comes before the action
#pragma unroll
      for (int r = 0; r < R; ++r) m[r] = 0u;
      pl = 0; passed = 0u; done = 0u;
    }
    if (!actions) {   // GoEnv.uniform_random_action (:78-81) on the h = 1 lanes: the k-th valid point, k == n: the pass
      uint32_t v[R], p[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        v[r] = full & ~m[r];
        p[r] = (uint32_t)__popc(v[r]) + (r ? p[r - 1] : 0u);
      }
      const uint32_t n = p[R - 1];
      const uint64_t u = splitmix_next(x);
      const uint32_t k = (uint32_t)(((u >> 32) * (uint64_t)(n + 1)) >> 32);
      int rr;
      uint32_t pos;
      kth_set_bit<R>(v, p, k < n ? k : 0u, rr, pos);
      a = k < n ? rr * N + (int)pos : P;
      const int ah = (int)dpp0<QP_HALF>((uint32_t)a);
      a = h ? a : ah;
      if (on && !frozen && (lane & 3) == 2) rng[b] = x;
    }
    const bool in_range = a >= 0 && a <= P;
    const bool is_pass = a == P;
    bool bad = !in_range || frozen;
    {
      int tr = 0, tc = 0;
      if (in_range && !is_pass) split_action(a, N, inv16, tr, tc);
      const uint32_t oh = (in_range && !is_pass) ? (1u << tr) : 0u;
      uint32_t hit = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) hit |= (uint32_t)__builtin_amdgcn_sbfe((int)oh, r, 1) & m[r];
      uint32_t occ = (hit >> tc) & 1u;                    // (meaningful on the h = 1 lanes: plane 3 at the point)
      const uint32_t occh = dpp0<QP_HALF>(occ);
      occ = h ? occ : occh;
      bad = bad || occ != 0u;
    }
    // the mask of a board that does not move stays what it was (zero after a reset): parked in the result rows now
    WAVE_SYNC();
    if (h == 1 && c == 0) {
      uint32_t *pi = rows + 2 * PL + bl * RS;
#pragma unroll
      for (int r = 0; r < R; ++r) pi[r] = m[r];
    }
    // ... and the h = 1 lanes take their colour's stones over from the h = 0 lanes
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t t = dpp0<QP_HALF>(m[r]);
      m[r] = B3(hm, t, m[r], T_SEL);
    }
    const bool moving = !bad && !is_pass;
    const bool mine = c == pl;
    int ar = 0, ac = 0;
    if (moving) split_action(a, N, inv16, ar, ac);
    const uint32_t bit = moving ? (1u << ac) : 0u;
    const uint32_t onehot = moving ? (1u << ar) : 0u;
    bool boxed;
    {
      uint32_t Q[R], dq[R], acc = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        Q[r] = (uint32_t)__builtin_amdgcn_sbfe((int)onehot, r, 1) & bit;
        if (mine) m[r] |= Q[r];
      }
      dilate_regs<R>(Q, dq);
#pragma unroll
      for (int r = 0; r < R; ++r) acc |= dq[r] & full & ~m[r];
      asm volatile("" : "+v"(acc));   // (pinned before the passes, see k_next_states16)
      boxed = acc == 0u;
    }
    const int nbrd = (int)((B - b_first) < kNB16 ? (B - b_first) : kNB16);
    // ---------------------------------------------------------------- six passes: class 2 j + h of all 32 units at once
    uint32_t alive[R], multi[R];
    {
      uint32_t c0[R], c1[R], c2[R], mrev[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        c0[r] = c1[r] = c2[r] = 0u;
        mrev[r] = __brev(m[r]);
      }
#pragma unroll 1
      for (int j = 0; j < (kCwClasses + 1) / 2; ++j) {
        uint32_t f[R], g[R];
        {
          uint32_t ee[R + 1];
          const uint4 *pc = reinterpret_cast<const uint4 *>(cwt + (2 * j + h) * 20);
#pragma unroll
          for (int i = 0; i < (R + 3) / 4; ++i) {
            const uint4 d = pc[i];
            if (4 * i < R) ee[4 * i] = d.x;
            if (4 * i + 1 < R) ee[4 * i + 1] = d.y;
            if (4 * i + 2 < R) ee[4 * i + 2] = d.z;
            if (4 * i + 3 < R) ee[4 * i + 3] = d.w;
          }
#pragma unroll
          for (int r = 0; r < R; ++r) ee[r] = B3(ee[r], m[r], dpp0<QP_COLOUR>(m[r]), TA & ~(TB | TC) & 0xFF) & full;
          ee[R] = 0;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const uint32_t xx = r > 0 ? B3(shl1(ee[r]), ee[r] >> 1, ee[r - 1], T_OR3) : (shl1(ee[r]) | (ee[r] >> 1));
            f[r] = B3(m[r], xx, ee[r + 1], T_AND_OR2);
          }
        }
        flood2_serial_regs<R>(m, mrev, f, g);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint32_t k0 = c0[r] & g[r];
          c0[r] ^= g[r];
          const uint32_t k1 = c1[r] & k0;
          c1[r] ^= k0;
          c2[r] |= k1;
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t p0 = dpp0<QP_HALF>(c0[r]), p1 = dpp0<QP_HALF>(c1[r]), p2 = dpp0<QP_HALF>(c2[r]);
        const uint32_t k0 = c0[r] & p0;
        const uint32_t s1 = B3(c1[r], p1, k0, TA ^ TB ^ TC);
        const uint32_t k1 = B3(c1[r], p1, k0, (TA & TB) | (TA & TC) | (TB & TC));
        const uint32_t s2 = B3(c2[r], p2, k1, TA ^ TB ^ TC);
        const uint32_t s3 = B3(c2[r], p2, k1, (TA & TB) | (TA & TC) | (TB & TC));
        alive[r] = B3(c0[r], c1[r], c2[r], T_OR3) | B3(p0, p1, p2, T_OR3);
        multi[r] = B3(s3, s2, s1, T_OR_AND);
      }
    }
    // ---------------------------------------------------------------- captures, ko, class patch (as k_next_states16)
    int ko_r = -1;
    uint32_t ko_bit = 0;
    uint32_t ndead = 0;
    {
      uint32_t dead[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        dead[r] = (!mine && moving) ? (m[r] & ~alive[r]) : 0u;
        ndead += (uint32_t)__popc(dead[r]);
      }
      if (__ballot(ndead != 0u)) {
        uint32_t krow = 0, kcols = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          krow += dead[r] ? (uint32_t)r : 0u;
          kcols |= dead[r];
          m[r] &= ~dead[r];
        }
        if (ndead == 1u && boxed) { ko_r = (int)krow; ko_bit = kcols; }
        uint32_t deadp[R];
#pragma unroll
        for (int r = 0; r < R; ++r) deadp[r] = dpp0<QP_COLOUR>(dead[r]);
        {
          uint32_t G0[R], t[R], n0 = 0;
#pragma unroll
          for (int r = 0; r < R; ++r) G0[r] = mine ? (m[r] & ~alive[r]) : 0u;
          dilate_regs<R>(G0, t);
#pragma unroll
          for (int r = 0; r < R; ++r) n0 += (uint32_t)__popc(t[r] & deadp[r]);
          const uint32_t g0m = n0 >= 2u ? ~0u : 0u;
#pragma unroll
          for (int r = 0; r < R; ++r) multi[r] |= G0[r] & g0m;
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
    // ---------------------------------------------------------------- the new flags (go_env.py:56-64 / gogame.py:77-87)
    if (!bad) {
      if (is_pass) { if (passed) done = 1u; passed = 1u; } else passed = 0u;
      pl ^= 1;
    }
    // ---------------------------------------------------------------- out: rows to LDS; the next mover's mask on ITS lanes
    // (for a board that moved: after the move the lanes of the mover's colour are the opponent's of the next mover)
    WAVE_SYNC();
    {
      uint32_t xs[R], e2[R], nb[R];
      const uint32_t kohot = ko_r >= 0 ? (1u << ko_r) : 0u;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t pm = dpp0<QP_COLOUR>(m[r]), pmulti = dpp0<QP_COLOUR>(multi[r]);
        e2[r] = full & ~(m[r] | pm);
        xs[r] = B3(e2[r], m[r] & multi[r], pm & ~pmulti, T_OR3);
      }
      dilate_regs<R>(xs, nb);
      if (h == 0) {
        uint32_t *const pr = rows + c * PL + bl * RS;
#pragma unroll
        for (int r = 0; r < R; ++r) pr[r] = m[r];
        if (!bad && !mine) {   // the next mover's lane owns the mask of a board that moved
          uint32_t *pi = rows + 2 * PL + bl * RS;
#pragma unroll
          for (int r = 0; r < R; ++r)
            pi[r] = (full & ~(e2[r] & nb[r])) | ((uint32_t)__builtin_amdgcn_sbfe((int)kohot, r, 1) & ko_bit);
        }
        if (c == 0) rows[bl * RS + R] = (uint32_t)pl | (passed << 1) | (done << 2);
      }
    }
    // ---------------------------------------------------------------- GoEnv.reward: Tromp-Taylor areas when they matter
    int area_own = 0, area_oth = 0;
    if (rewards && __ballot(on && (HEUR || done != 0u))) {
      uint32_t e[R], erev[R], f[R], g[R], d[R];
      dilate_regs<R>(m, d);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        e[r] = full & ~(m[r] | dpp0<QP_COLOUR>(m[r]));
        erev[r] = __brev(e[r]);
        f[r] = e[r] & d[r];                                // empty points next to this lane's colour
      }
      flood2_serial_regs<R>(e, erev, f, g);                // ... and every empty point connected to them
      uint32_t cnt = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) cnt += (uint32_t)__popc(m[r]) + (uint32_t)__popc(g[r] & ~dpp0<QP_COLOUR>(g[r]));
      area_own = (int)cnt;
      area_oth = (int)dpp0<QP_COLOUR>(cnt);
    }
    if (on && (lane & 3) == 0) {   // (c = 0: area_own is black's)
      const float margin = (float)(area_own - area_oth) - komi;
      float rwd;
      if (HEUR) rwd = done ? (margin > 0.f ? 1.f : -1.f) * (float)P : margin;
      else rwd = done ? (margin > 0.f ? 1.f : margin < 0.f ? -1.f : 0.f) : 0.f;
      if (rewards) rewards[b] = rwd;
      if (dones) dones[b] = (uint8_t)done;
      if (status) status[b] = bad ? GG_STATUS_ILLEGAL : GG_STATUS_OK;
      if (taken) taken[b] = a;
      // (a one-ply gg_batch_rollout: the last action of a game that did not move is -1, the plies played are counted)
      if (last_actions) last_actions[b] = bad ? -1 : a;
      if (steps_done && !bad) atomicAdd(reinterpret_cast<unsigned long long *>(steps_done) + b, 1ull);
    }
    WAVE_SYNC();
    emit_rows16<R>(states + b_first * (int64_t)S, nbrd, rows, PL, RS, lds + Lds16<R>::kGrpBits, lut, lane);
    WAVE_SYNC();
  }
}

}  // namespace gg
