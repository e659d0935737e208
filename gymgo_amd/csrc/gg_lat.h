// gg_lat.h - the LATENCY-SHAPED multi-ply kernel for batches that leave the machine under-filled (config 2: 4 096 games of
// 9x9 are four boards per SIMD): ONE ROW PER LANE, one board per DPP row of 16 lanes (R <= 13) or per 32 lanes (R = 19),
// the whole ply in registers - no LDS between the phases, no layout change, no transposes.
//
// The other two families pay per ply for work that only amortises over many boards: gg_v2.h re-derives every liberty class
// (22 floods per board, flood-per-lane layout reached through LDS), gg_v4.h keeps the classes but hands the board from the
// quad layout to the flood lanes and back through LDS three times per ply.  With four boards per SIMD neither has the
// boards to hide those round trips behind: the ply is a serial dependency chain (1.46 us per ply for two boards in
// k_rollout2, 2.2 us for a lone 9x9 wave of k_rollout4).  Here the chain itself is short:
//   * lane (b, r) holds row r of board b: the mover's and the opponent's stones, M = the stones of either colour whose group
//     has >= 2 liberties (every other stone is in atari), the mover's invalid-move mask, the flag word and the generator;
//   * the draw: popcount of the lane's valid points, prefix / total inside the board by DPP row shifts / rotations, every
//     lane draws the same number, the lane that holds the k-th valid point finds the bit (binary search on popcounts);
//   * a move at q changes the classes of the groups ADJACENT to q only (gg_v4.h): the opponent's group at each of q's four
//     neighbours and the mover's group G that the stone joins - FIVE floods, run AT THE SAME TIME in bit fields of the same
//     registers: a 9x9 row is 9 bits, so three floods share a VGPR (10-bit fields, one guard bit; 13x13: two 16-bit fields;
//     19x19: one).  A flood step is "two rows up / down (DPP moves) + a complete horizontal run fill (carry chain, once per
//     direction - gg_common.h)", the closure test is the head of the next step;
//   * liberties = dilate & empty per field, counted per lane, summed per board by four DPP rotations (all five counts, the
//     number of captured stones and "q is boxed in" travel in two words); the class patch and the next mover's mask follow
//     point-wise exactly as in gg_v4.h (phase 3) - here on one row per lane with the rows above / below one DPP move away.
// 296 VALU + 17 SALU wave-instructions per ply for FOUR boards (PMC, 9x9), none of them an LDS or memory instruction.  A lone
// wave issues one instruction per ~4 cycles whether or not it depends on the previous one, so at four boards per SIMD the
// launch costs instructions per wave-ply x that, and a second / third wave per SIMD fills the slots the first leaves
// (8 192 games of 9x9: x1.75).  Config 2 (4 096 games of 9x9 x 256 plies): 2.40e9 (k_rollout2) -> 4.5e9 env steps/s.
// Byte planes enter through LDS once per launch (aligned staging as everywhere, gg_common.h); their first classes come from
// the constant-weight analysis of gg_v2.h restated in this layout (eleven floods in lock-step, black and white in two fields
// of one register) - which is why byte-plane launches pay from 3 / 4 / 64 plies on.  TRACKED boards (IO = 2) carry their
// classes: a lane loads and stores its own five row words, no LDS at all, and the kernel pays from ONE ply per launch
// (3.5 us per ply as a hipGraph at config 2's size).  Take-over points: gg_kernels.hip, use_lat.
// tests/devtools/lat_model.py is a lane-by-lane NumPy model of this file (same DPP shuffles, same bit tricks) checked against
// oracle/gg_oracle.c on the CPU.  Reference: the loop gym_go/envs/go_env.py:49-81 (uniform_random_action + step) over
// gym_go/gogame.py:34-87, gym_go/state_utils.py:24-83,159-180.
#pragma once
#include "gg_v2.h"
#include "gg_v4.h"   // GG_PROF phase clocks (A/B builds only)

namespace gg {

template <int R>
struct Lat {
  static constexpr int LPB = R <= 15 ? 16 : 32;                   // lanes per board
  static constexpr int NBW = kWave / LPB;                         // boards per wave: 4 / 2
  static constexpr int FW = R <= 9 ? 10 : (R <= 15 ? 16 : 32);    // bits per flood field (one guard bit above the row unless the field is the register)
  static constexpr int NF = 32 / FW;                              // fields per register: 3 / 2 / 1
  static constexpr int NFL = 5;                                   // floods per ply: up, down, left, right (opponent), G (mover)
  static constexpr int NREG = (NFL + NF - 1) / NF;                // 2 / 3 / 5 (the five-flood ply: lat_play_full)
  static constexpr int NFL3 = 3;                                  // floods of the usual ply (lat_play): two opponent slots + G
  static constexpr int NREG3 = (NFL3 + NF - 1) / NF;              // 1 / 2 / 3
  static constexpr uint32_t FM = FW >= 32 ? 0xFFFFFFFFu : ((1u << (FW & 31)) - 1u);
  static constexpr int kBits = R <= 16 ? 16 : 32;                 // width of the k-th-set-bit search
  static constexpr bool kSat = R > 9;                             // per-lane liberty counts saturated at 2 (the board sums are 8-bit fields)
  static constexpr int kIoBytes = Cfg<R>::kIoBytes;
  static constexpr int kBsWords = ((((15 + 6 * R * R + 31) / 32 + 1) + LPB - 1) / LPB) * LPB;   // bit-string of one board, a multiple of LPB
  static_assert(R < LPB, "rows >= N of a board are zero: nothing leaks between the boards of a wave");
  static_assert(NF == 1 || R < FW, "guard bit between the fields");
};

// rows above / below: one DPP move.  16 lanes per board: row shifts (zero fill at the row's ends); 32: wave shifts (rows
// >= N of every board are zero, so nothing crosses a board boundary)
template <int LPB> __device__ __forceinline__ uint32_t lat_above(uint32_t x) { return LPB == 16 ? dpp0<0x111>(x) : dpp0<0x138>(x); }   // lane i reads lane i - 1
template <int LPB> __device__ __forceinline__ uint32_t lat_below(uint32_t x) { return LPB == 16 ? dpp0<0x101>(x) : dpp0<0x130>(x); }   // lane i reads lane i + 1

// 4-neighbourhood dilation of a (possibly field-packed) row set; the centre is not part of the result.  Bits that cross a
// field's edge land on a guard bit / beyond N: every caller masks with a set that is zero there.
template <int LPB> __device__ __forceinline__ uint32_t lat_dilate(uint32_t x) {
  return B3(shl1(x), x >> 1, lat_above<LPB>(x), T_OR3) | lat_below<LPB>(x);
}

// sum over the lanes of a board, result in every lane: four row rotations (+ one swap of the two rows of a 32-lane board)
template <int LPB> __device__ __forceinline__ uint32_t lat_board_sum(uint32_t x) {
  x += dpp0<0x128>(x);   // row_ror:8
  x += dpp0<0x124>(x);
  x += dpp0<0x122>(x);
  x += dpp0<0x121>(x);
  if (LPB == 32) x += (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x401F);   // lane ^ 16 (and 0x1F, or 0, xor 0x10)
  return x;
}
template <int LPB> __device__ __forceinline__ int lat_board_max(int x) {
  int y;
  y = (int)dpp0<0x128>((uint32_t)x); x = x > y ? x : y;
  y = (int)dpp0<0x124>((uint32_t)x); x = x > y ? x : y;
  y = (int)dpp0<0x122>((uint32_t)x); x = x > y ? x : y;
  y = (int)dpp0<0x121>((uint32_t)x); x = x > y ? x : y;
  if (LPB == 32) { y = __builtin_amdgcn_ds_swizzle(x, 0x401F); x = x > y ? x : y; }
  return x;
}
// inclusive prefix sum over the lanes of a board (half_scan of gg_v2.h without / with the row broadcast)
template <int LPB> __device__ __forceinline__ uint32_t lat_board_scan(uint32_t v) {
  v += dpp0<0x111>(v);
  v += dpp0<0x112>(v);
  v += dpp0<0x114>(v);
  v += dpp0<0x118>(v);
  if (LPB == 32) v += dpp0<0x142, 0xA>(v);   // lane 15 of rows 0 / 2 added to every lane of rows 1 / 3
  return v;
}

// position of the tt-th (0-based) set bit of v (tt < popc(v); anything otherwise): binary search on popcounts, branch-free
// on 0 / ~0 masks (kth_set_bit of gg_v4.h for one row)
template <int BITS> __device__ __forceinline__ uint32_t lat_kth_bit(uint32_t v, uint32_t tt) {
  // five instructions per level (a lone wave pays per instruction, whatever it is): the field by v_bfe, its popcount added to
  // ~tt by the same v_bcnt (c + ~tt = c - tt - 1 is negative iff tt >= c: the bit is in the upper half - and is ~(tt - c) then)
  uint32_t ps = 0, ntt = ~tt;
#pragma unroll
  for (int sh = BITS / 2; sh >= 1; sh >>= 1) {
    const uint32_t e = (uint32_t)__popc(__builtin_amdgcn_ubfe(v, ps, (uint32_t)sh)) + ntt;
    const uint32_t ge = (uint32_t)((int32_t)e >> 31);
    ntt = B3(ge, e, ntt, T_SEL);
    ps = B3((uint32_t)sh, ge, ps, T_ANDOR);                             // ps | (sh & ge)
  }
  return ps;
}

// one visit of a row: the seeds s (a subset of ma) fill their runs towards the MSB, the row is flipped and fills towards
// the MSB again; the result is in the OTHER bit order (ma / mb = the mask in the current / the other order): FLOOD_VISIT
// of gg_common.h without the vertical term.  Works on field-packed rows as long as a zero bit separates the fields.
__device__ __forceinline__ uint32_t lat_visit(uint32_t ma, uint32_t mb, uint32_t s) {
  const uint32_t t = ma + s;
  const uint32_t u = B3(t, s, ma, T_SEL);
  const uint32_t v = __brev(u);
  const uint32_t t2 = mb + v;
  return B3(t2, v, mb, T_SEL);
}

// K row sets flooded in lock-step to their fixed points (Jacobi over the rows; complete horizontal fill per step), F = seeds
// (subsets of Mk) in, filled sets out, Mkr = Mk bit-reversed.  A step moves the fill TWO rows up / down before it fills the
// runs: the second move costs three instructions per register and takes the mean number of steps from 4.4 to 2.7 on 9x9
// (6.3 -> 3.7 on 13x13, 5.9 -> 3.4 on 19x19; a third move buys 0.4 - 0.7 more for as many instructions as it saves:
// tests/devtools/lat_model.py) - random-play groups are blobs, and a blob only needs the vertical moves.  The bit order
// alternates from step to step (one v_bfrev per visit); the closure test of a step - "a fillable point above / below a filled
// one" - is the head of the next step, so a flood that is already closed costs 3 instructions + the branch.
template <int LPB, int K>
__device__ __forceinline__ void lat_flood(uint32_t (&F)[K], const uint32_t (&Mk)[K], const uint32_t (&Mkr)[K]) {
#pragma unroll
  for (int k = 0; k < K; ++k) F[k] = lat_visit(Mk[k], Mkr[k], F[k]);   // the seeds' own runs; now bit-reversed
#pragma unroll 1
  for (int it = 0; it < 512; ++it) {
    uint32_t open = 0, s[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t v = lat_above<LPB>(F[k]) | lat_below<LPB>(F[k]);
      const uint32_t o = B3(v, Mkr[k], F[k], T_AND_ANDN);
      s[k] = o | F[k];
      open |= o;
    }
    if (__ballot(open != 0) == 0) {
#pragma unroll
      for (int k = 0; k < K; ++k) F[k] = __brev(F[k]);
      return;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      s[k] = B3(lat_above<LPB>(s[k]) | lat_below<LPB>(s[k]), Mkr[k], s[k], T_ANDOR);   // the second row up / down
      F[k] = lat_visit(Mkr[k], Mk[k], s[k]);                                          // back in normal order
    }
    open = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t v = lat_above<LPB>(F[k]) | lat_below<LPB>(F[k]);
      const uint32_t o = B3(v, Mk[k], F[k], T_AND_ANDN);
      s[k] = o | F[k];
      open |= o;
    }
    if (__ballot(open != 0) == 0) return;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      s[k] = B3(lat_above<LPB>(s[k]) | lat_below<LPB>(s[k]), Mk[k], s[k], T_ANDOR);
      F[k] = lat_visit(Mk[k], Mkr[k], s[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) F[k] = __brev(F[k]);   // (iteration bound: cannot be reached for R <= 19)
}

// The stones (of either colour) whose group has >= 2 liberties, from the stones alone: the constant-weight code of gg_v2.h
// (point q gets the q-th 11-bit word of weight 5; flood i is seeded next to the empty points whose word has bit i; a group
// with one liberty is reached by exactly 5 floods, with two or more by >= 6) in the row-per-lane layout - eleven floods in
// lock-step, black and white in two fields of a register (19x19: two passes).  cwm[i] = kCw.m[i][the lane's row], read by the
// caller together with the boards (one round trip for all of the launch's first reads: cw_table_issue, gg_v2.h).
template <int R>
__device__ __forceinline__ uint32_t lat_classes(uint32_t bl, uint32_t wh, uint32_t full, const uint32_t (&cwm)[kCwClasses]) {
  using L = Lat<R>;
  constexpr int NC = L::NF >= 2 ? 1 : 2;
  const uint32_t E = full & ~(bl | wh);
  uint32_t d[kCwClasses];
#pragma unroll
  for (int i = 0; i < kCwClasses; ++i) d[i] = lat_dilate<L::LPB>(E & cwm[i]);
  uint32_t multi_all = 0;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const uint32_t P = NC == 1 ? (bl | (wh << (L::FW & 31))) : (c ? wh : bl);
    const uint32_t Pr = __brev(P);
    uint32_t F[kCwClasses], Mk[kCwClasses], Mkr[kCwClasses];
#pragma unroll
    for (int i = 0; i < kCwClasses; ++i) {
      F[i] = (NC == 1 ? (d[i] | (d[i] << (L::FW & 31))) : d[i]) & P;
      Mk[i] = P;
      Mkr[i] = Pr;
    }
    lat_flood<L::LPB, kCwClasses>(F, Mk, Mkr);
    uint32_t alive, multi;
    classify11(F, alive, multi);
    multi_all |= NC == 1 ? ((multi & L::FM) | (multi >> (L::FW & 31))) : multi;
  }
  return multi_all & full;
}

// Board emission, L1 rows -> HBM, by the lanes of ONE board (emit_store_h of gg_v2.h for a board of LPB lanes): the six
// planes are ORed row by row into a bit-string (bit (g & 15) + i = board byte i), every aligned 16-byte vector of HBM is one
// halfword of it, expanded through the 8 bits -> 8 bytes table; the ragged head / tail leave as single bytes.
template <int R>
__device__ __forceinline__ void lat_emit(uint8_t *g, uint32_t black, uint32_t white, uint32_t invalid, uint32_t turn,
                                         uint32_t passed, uint32_t done, uint32_t full, int N, int r, uint32_t *bs,
                                         const uint2 *lut, bool wr) {
  using L = Lat<R>;
  const int P = N * N, S = 6 * P;
  const uint32_t mo = (uint32_t)((uintptr_t)g & 15u);
  WAVE_SYNC();
#pragma unroll
  for (int k = 0; k < L::kBsWords / L::LPB; ++k) bs[r + L::LPB * k] = 0;
  WAVE_SYNC();
  if (wr && r < N) {
    const uint32_t rows[6] = {black, white, turn ? full : 0u, invalid, passed ? full : 0u, done ? full : 0u};
    const uint32_t q0 = mo + (uint32_t)(r * N);
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      if (rows[p]) {
        const uint32_t q = q0 + (uint32_t)(p * P);
        const uint64_t x = (uint64_t)rows[p] << (q & 31u);
        uint32_t *w = bs + (q >> 5);
        atomicOr(w, (uint32_t)x);
        if ((uint32_t)(x >> 32)) atomicOr(w + 1, (uint32_t)(x >> 32));
      }
    }
  }
  WAVE_SYNC();
  if (wr) {
    uint8_t *ga = g - mo;
    const int end = (int)mo + S;
    const int v0 = mo ? 1 : 0, v1 = end >> 4;
    const uint8_t *bb = reinterpret_cast<const uint8_t *>(bs);
    for (int v = v0 + r; v < v1; v += L::LPB) {
      const uint2 lo = lut[bb[2 * v]], hi = lut[bb[2 * v + 1]];
      V16a o;
      o.w[0] = lo.x; o.w[1] = lo.y; o.w[2] = hi.x; o.w[3] = hi.y;
      *reinterpret_cast<V16a *>(ga + 16 * v) = o;
    }
    if (v1 >= v0) {
      const int head = mo ? 16 - (int)mo : 0, tail = end & 15;
#pragma unroll
      for (int e = 0; e < 2; ++e) {   // (a board of 16 lanes takes the two ragged edges one after the other)
        int j = -1;
        if (e == 0) { if (r < head) j = r; }
        else if (r < tail) j = S - tail + r;
        if (j >= 0) {
          const uint32_t q = mo + (uint32_t)j;
          g[j] = (uint8_t)((bs[q >> 5] >> (q & 31u)) & 1u);
        }
      }
    } else {   // the board lies inside one 16-byte chunk (N = 2 only: 24 bytes never do, kept for symmetry)
      for (int i = r; i < S; i += L::LPB) {
        const uint32_t q = mo + (uint32_t)i;
        g[i] = (uint8_t)((bs[q >> 5] >> (q & 31u)) & 1u);
      }
    }
  }
}

// Phases 2 - 6 of a ply on the boards of the wave: Q = the new stone (one bit in the lane of its row, 0 in every other lane and
// on a board that passes or does not move), pass (the same in every lane of a board), lv = 0 / ~0: the board moves this ply.
// me / op / M / inv / fl are updated in place (roles swapped for the boards that moved).
// FIVE floods (round 5's ply): one per neighbour of q + G.  Since round 6 the rare path of lat_play below.
template <int R>
__device__ __forceinline__ void lat_play_full(uint32_t &me, uint32_t &op, uint32_t &M, uint32_t &inv, uint32_t &fl, const uint32_t Q,
                                           const bool pass, const uint32_t lv, const uint32_t full) {
  using L = Lat<R>;
  constexpr int LPB = L::LPB, FW = L::FW, NF = L::NF, NREG = L::NREG;
  // 2. the stone, its four neighbours, the five floods
  const uint32_t me1 = me | Q;
  const uint32_t su = lat_below<LPB>(Q), sd = lat_above<LPB>(Q);   // the point above q lies one row up: that lane takes Q from the lane below it
  const uint32_t sl = Q >> 1, sr = shl1(Q);
  const uint32_t open = B3(B3(su, sd, sl, T_OR3) | sr, full, op, T_AND_ANDN);   // on-board neighbours of q that do not hold an opponent stone
  uint32_t F[NREG], Mk[NREG], Mkr[NREG];
  {
    const uint32_t seeds[L::NFL] = {su & op, sd & op, sl & op, sr & op, Q};
#pragma unroll
    for (int k2 = 0; k2 < NREG; ++k2) { F[k2] = 0; Mk[k2] = 0; }
#pragma unroll
    for (int f = 0; f < L::NFL; ++f) {
      F[f / NF] |= seeds[f] << ((FW * (f % NF)) & 31);
      Mk[f / NF] |= (f < 4 ? op : me1) << ((FW * (f % NF)) & 31);
    }
#pragma unroll
    for (int k2 = 0; k2 < NREG; ++k2) Mkr[k2] = __brev(Mk[k2]);
  }
  lat_flood<LPB, NREG>(F, Mk, Mkr);
  uint32_t fr[L::NFL];
#pragma unroll
  for (int f = 0; f < L::NFL; ++f) fr[f] = NF == 1 ? F[f] : ((F[f / NF] >> ((FW * (f % NF)) & 31)) & L::FM);
  const uint32_t U = B3(fr[0], fr[1], fr[2], T_OR3) | fr[3], G = fr[4];
  const uint32_t C = U & ~M;                          // an opponent group next to q that was in atari: captured
  // 3. liberties of the five groups: dilate & empty (G's include the captured points), counted per lane, summed per board
  const uint32_t E1 = full & ~(me1 | op), EG = E1 | C;
  uint32_t W1 = 0, W2 = 0;
  {
    uint32_t Ee[NREG];
#pragma unroll
    for (int k2 = 0; k2 < NREG; ++k2) Ee[k2] = 0;
#pragma unroll
    for (int f = 0; f < L::NFL; ++f) Ee[f / NF] |= (f < 4 ? E1 : EG) << ((FW * (f % NF)) & 31);
    uint32_t Lb[NREG];
#pragma unroll
    for (int k2 = 0; k2 < NREG; ++k2) Lb[k2] = lat_dilate<LPB>(F[k2]) & Ee[k2];
#pragma unroll
    for (int f = 0; f < L::NFL; ++f) {
      uint32_t c = (uint32_t)__popc(NF == 1 ? Lb[f] : (Lb[f / NF] & (L::FM << ((FW * (f % NF)) & 31))));
      if (L::kSat) c = c < 2u ? c : 2u;
      if (f < 4) W1 |= c << (8 * f);
      else W2 = c;
    }
    uint32_t pc = (uint32_t)__popc(C);
    pc = pc < 2u ? pc : 2u;
    W2 |= (pc << 8) | (open ? 0x10000u : 0u);
  }
  const uint32_t S1 = lat_board_sum<LPB>(W1), S2 = lat_board_sum<LPB>(W2);
  // 4. class patch: every flooded group leaves M and comes back with >= 2 liberties (count + 126 carries into bit 7)
  const uint32_t g1 = S1 + 0x7E7E7E7Eu;
  uint32_t M1 = B3(M, U, G, TA & ~(TB | TC) & 0xFF);
#pragma unroll
  for (int f = 0; f < 4; ++f) M1 = B3((uint32_t)((int32_t)(g1 << (24 - 8 * f)) >> 31), fr[f], M1, T_ANDOR);
  M1 = B3((uint32_t)((int32_t)(1u - (S2 & 0xFFu)) >> 31), G, M1, T_ANDOR);
  uint32_t K = 0;
  if (__ballot(C != 0)) {   // some board of the wave captures
    // ko: exactly one stone died and q is boxed in (gym_go/gogame.py:72-75, state_utils.adj_data)
    if (((S2 >> 8) & 0xFFu) == 1u && ((S2 >> 16) & 0xFFu) == 0u) K = C;
    // a mover's group in atari next to a captured stone (not G: its count above includes them) gains a liberty
    uint32_t X[1], Xm[1], Xr[1];
    Xm[0] = B3(me1, M, G, TA & ~(TB | TC) & 0xFF);
    X[0] = lat_dilate<LPB>(C) & Xm[0];
    if (__ballot(X[0] != 0)) {
      Xr[0] = __brev(Xm[0]);
      lat_flood<LPB, 1>(X, Xm, Xr);
      M1 |= X[0];
    }
  }
  // 5. the next mover's invalid-move mask (state_utils.compute_invalid_moves restated point-wise): an empty
  // point is playable iff some neighbour is empty, a next-mover stone with >= 2 liberties or a mover's stone in atari
  const uint32_t op2 = op & ~C;
  const uint32_t E2 = full & ~(me1 | op2);
  const uint32_t xs = E2 | B3(M1, op2, me1, T_SEL);
  const uint32_t nbs = lat_dilate<LPB>(xs);
  const uint32_t inv2 = B3(full, E2, nbs, TA & ~(TB & TC) & 0xFF) | K;
  // 6. roles swap for the boards that moved; flags: turn flips, passed = pass, done = two passes in a row
  me = B3(lv, op2, me1, T_SEL);
  op = B3(lv, me1, op2, T_SEL);
  inv = B3(lv, inv2, inv, T_SEL);
  M = M1;
  const uint32_t pm = pass ? ~0u : 0u;
  const uint32_t fl2 = ((fl ^ 1u) & 1u) | (pm & 2u) | (pm & (fl << 1) & 4u);
  fl = B3(lv, fl2, fl, T_SEL);
}

// The usual ply (round 6): THREE floods.  An opponent group next to q that is NOT in M had q as its only liberty - it is
// captured whatever its shape - so only the neighbours of q that are in M (>= 2 liberties before the move) need a liberty
// count of their own, and more than two DISTINCT such groups next to one point are rare (1.2 % of the moves, measured on
// stationary 9x9 and 19x19 positions; two: 10 - 13 %).  Slot A floods every atari neighbour + the FIRST neighbour in M
// (priority up, down, left, right), slot B the SECOND one, the third field is G: a 9x9 ply is ONE flood register instead of
// two, 13x13 two instead of three, 19x19 three instead of five, and all the board sums travel in one word.  Neighbours in M
// beyond the second must lie in A | B (they are stones of the same groups); when one does not, a second round floods the
// left-over neighbours (4 % of the wave-plies at four 9x9 boards per wave).  tests/devtools/lat_model.py: the same in NumPy.
template <int R>
__device__ __forceinline__ void lat_play(uint32_t &me, uint32_t &op, uint32_t &M, uint32_t &inv, uint32_t &fl, const uint32_t Q,
                                         const bool pass, const uint32_t lv, const uint32_t full) {
  using L = Lat<R>;
  constexpr int LPB = L::LPB, FW = L::FW, NF = L::NF, NREG = L::NREG3, NFL = L::NFL3;
  // 2. the stone, its four neighbours, the three floods
  const uint32_t me1 = me | Q;
  const uint32_t su = lat_below<LPB>(Q), sd = lat_above<LPB>(Q);   // the point above q lies one row up: that lane takes Q from the lane below it
  const uint32_t sl = Q >> 1, sr = shl1(Q);
  const uint32_t nbr = B3(su, sd, sl, T_OR3) | sr;
  const uint32_t open = B3(nbr, full, op, T_AND_ANDN);   // on-board neighbours of q that do not hold an opponent stone
  const uint32_t mAll = B3(nbr, op, M, TA & TB & TC);    // the opponent's stones next to q whose group had >= 2 liberties
  const uint32_t aAll = B3(nbr, op, M, T_AND_ANDN);      // ... and the ones in atari: captured
  uint32_t seedA, seedB;
  {
    const uint32_t mUl = su & mAll, mDl = sd & mAll, mL = sl & mAll, mR = sr & mAll;   // in the lanes of their rows
    const uint32_t fU = lat_above<LPB>(mUl), fD = lat_below<LPB>(mDl);                 // ... and as flags at q's own bit, in q's lane
    const uint32_t gU = lat_above<LPB>(fU);                                            // the lane below q learns about the point above q
    const uint32_t fL = shl1(mL), fR = mR >> 1;
    const uint32_t p1L = B3(fL, fU, fD, TA & ~(TB | TC) & 0xFF);      // left is the first neighbour in M
    const uint32_t p2L = B3(fL, fU, fD, TA & (TB ^ TC));              // ... the second
    const uint32_t udl = B3(fU, fD, fL, T_OR3);
    const uint32_t p1R = fR & ~udl;
    const uint32_t p2R = fR & B3(fU, fD, fL, 0x16);                   // exactly one of up / down / left
    seedA = B3(aAll, mUl, B3(mDl, gU, p1L >> 1, (TA & ~TB & 0xFF) | TC), T_OR3) | shl1(p1R);
    seedB = B3(mDl & gU, p2L >> 1, shl1(p2R), T_OR3);
  }
  uint32_t F[NREG], Mk[NREG], Mkr[NREG];
  {
    const uint32_t seeds[NFL] = {seedA, seedB, Q};
#pragma unroll
    for (int k2 = 0; k2 < NREG; ++k2) { F[k2] = 0; Mk[k2] = 0; }
#pragma unroll
    for (int f = 0; f < NFL; ++f) {
      F[f / NF] |= seeds[f] << ((FW * (f % NF)) & 31);
      Mk[f / NF] |= (f < 2 ? op : me1) << ((FW * (f % NF)) & 31);
    }
#pragma unroll
    for (int k2 = 0; k2 < NREG; ++k2) Mkr[k2] = __brev(Mk[k2]);
  }
  lat_flood<LPB, NREG>(F, Mk, Mkr);
  uint32_t fr[NFL];
#pragma unroll
  for (int f = 0; f < NFL; ++f) fr[f] = NF == 1 ? F[f] : ((F[f / NF] >> ((FW * (f % NF)) & 31)) & L::FM);
  uint32_t U = fr[0] | fr[1];
  const uint32_t G = fr[2];
  const uint32_t C = fr[0] & ~M;                      // the opponent groups next to q that were in atari: captured
  // 3. liberties of the three groups: dilate & empty (G's include the captured points; the captured part of slot A has none
  // among the empty points - q was its only liberty), counted per lane, summed per board in ONE word: 8-bit fields A, B, G,
  // then the number of captured stones (6 bits) and "some on-board neighbour of q is not the opponent's" (2 bits)
  const uint32_t E1 = full & ~(me1 | op), EG = E1 | C;
  uint32_t W = 0;
  {
    uint32_t Ee[NREG];
#pragma unroll
    for (int k2 = 0; k2 < NREG; ++k2) Ee[k2] = 0;
#pragma unroll
    for (int f = 0; f < NFL; ++f) Ee[f / NF] |= (f < 2 ? E1 : EG) << ((FW * (f % NF)) & 31);
    uint32_t Lb[NREG];
#pragma unroll
    for (int k2 = 0; k2 < NREG; ++k2) Lb[k2] = lat_dilate<LPB>(F[k2]) & Ee[k2];
#pragma unroll
    for (int f = 0; f < NFL; ++f) {
      uint32_t c = (uint32_t)__popc(NF == 1 ? Lb[f] : (Lb[f / NF] & (L::FM << ((FW * (f % NF)) & 31))));
      if (L::kSat) c = c < 2u ? c : 2u;
      W |= c << (8 * f);
    }
    uint32_t pc = (uint32_t)__popc(C);
    pc = pc < 2u ? pc : 2u;
    W |= (pc << 24) | (open ? 0x40000000u : 0u);
  }
  const uint32_t S = lat_board_sum<LPB>(W);
  // ... and the rare second round: a neighbour of q in M that A | B do not cover is a THIRD (fourth) distinct group.  Up and
  // down are always first or second, so only the left and the right neighbour can be left over - each in a slot of its own:
  // one more flood of the two opponent fields (the masks of round one serve again; G's field stays empty), their liberties,
  // their board sums.  (4 % of the wave-plies at four 9x9 boards per wave, 3 % at two 19x19 boards.)
  uint32_t M1x = 0;   // round-two groups that keep >= 2 liberties
  if (__builtin_expect(__ballot((mAll & ~U) != 0u) != 0ull, 0)) {
    constexpr int K2 = NF == 1 ? 2 : 1;
    const uint32_t s2[2] = {(uint32_t)B3(sl, mAll, U, T_AND_ANDN), (uint32_t)B3(sr, mAll, U, T_AND_ANDN)};
    uint32_t F2[K2], Mk2[K2], Mkr2[K2];
#pragma unroll
    for (int k2 = 0; k2 < K2; ++k2) { Mk2[k2] = Mk[k2]; Mkr2[k2] = Mkr[k2]; }
    if (NF == 1) { F2[0] = s2[0]; F2[K2 - 1] = s2[1]; }
    else F2[0] = s2[0] | (s2[1] << (FW & 31));
    lat_flood<LPB, K2>(F2, Mk2, Mkr2);
    uint32_t Lb2[K2], W2 = 0, f2[2];
#pragma unroll
    for (int k2 = 0; k2 < K2; ++k2) Lb2[k2] = lat_dilate<LPB>(F2[k2]) & (NF == 1 ? E1 : (E1 | (E1 << (FW & 31))));
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      f2[f] = NF == 1 ? F2[f < K2 ? f : 0] : ((F2[0] >> ((FW * f) & 31)) & L::FM);
      uint32_t c = (uint32_t)__popc(NF == 1 ? Lb2[f < K2 ? f : 0] : (Lb2[0] & (L::FM << ((FW * f) & 31))));
      if (L::kSat) c = c < 2u ? c : 2u;
      W2 |= c << (8 * f);
    }
    const uint32_t g2 = lat_board_sum<LPB>(W2) + 0x00007E7Eu;
    M1x = ((uint32_t)((int32_t)(g2 << 24) >> 31) & f2[0]) | ((uint32_t)((int32_t)(g2 << 16) >> 31) & f2[1]);
    U |= f2[0] | f2[1];
  }
  // 4. class patch: every flooded group leaves M and comes back with >= 2 liberties (count + 126 carries into bit 7)
  const uint32_t g1 = S + 0x007E7E7Eu;
  uint32_t M1 = B3(M, U, G, TA & ~(TB | TC) & 0xFF) | M1x;
  M1 = B3((uint32_t)((int32_t)(g1 << 24) >> 31), fr[0] & M, M1, T_ANDOR);
  M1 = B3((uint32_t)((int32_t)(g1 << 16) >> 31), fr[1], M1, T_ANDOR);
  M1 = B3((uint32_t)((int32_t)(g1 << 8) >> 31), G, M1, T_ANDOR);
  uint32_t K = 0;
  if (__ballot(C != 0)) {   // some board of the wave captures
    // ko: exactly one stone died and q is boxed in (gym_go/gogame.py:72-75, state_utils.adj_data)
    if (((S >> 24) & 0x3Fu) == 1u && (S >> 30) == 0u) K = C;
    // a mover's group in atari next to a captured stone (not G: its count above includes them) gains a liberty
    uint32_t X[1], Xm[1], Xr[1];
    Xm[0] = B3(me1, M, G, TA & ~(TB | TC) & 0xFF);
    X[0] = lat_dilate<LPB>(C) & Xm[0];
    if (__ballot(X[0] != 0)) {
      Xr[0] = __brev(Xm[0]);
      lat_flood<LPB, 1>(X, Xm, Xr);
      M1 |= X[0];
    }
  }
  // 5. the next mover's invalid-move mask (state_utils.compute_invalid_moves restated point-wise): an empty
  // point is playable iff some neighbour is empty, a next-mover stone with >= 2 liberties or a mover's stone in atari
  const uint32_t op2 = op & ~C;
  const uint32_t E2 = full & ~(me1 | op2);
  const uint32_t xs = E2 | B3(M1, op2, me1, T_SEL);
  const uint32_t nbs = lat_dilate<LPB>(xs);
  const uint32_t inv2 = B3(full, E2, nbs, TA & ~(TB & TC) & 0xFF) | K;
  // 6. roles swap for the boards that moved; flags: turn flips, passed = pass, done = two passes in a row
  me = B3(lv, op2, me1, T_SEL);
  op = B3(lv, me1, op2, T_SEL);
  inv = B3(lv, inv2, inv, T_SEL);
  M = M1;
  const uint32_t pm = pass ? ~0u : 0u;
  const uint32_t fl2 = ((fl ^ 1u) & 1u) | (pm & 2u) | (pm & (fl << 1) & 4u);
  fl = B3(lv, fl2, fl, T_SEL);
}

template <int R>
struct LdsLat {
  using L = Lat<R>;
  static constexpr int kIo = 0;                                        // [NBW][kIoBytes] bytes: staged boards
  static constexpr int kBs = kIo + L::NBW * L::kIoBytes / 4;           // [NBW][kBsWords]: the emitters' bit-strings
  static constexpr int kTotal = kBs + L::NBW * L::kBsWords;
};

// gg_batch_rollout on byte planes (uint8 [B][6][N][N], in place): `plies` uniform-random plies per game, boards on-chip in
// between.  One single-wave workgroup per NBW boards.
// AUTO: auto_reset != 0 (every board of the batch is live for every ply of the launch: no liveness test, no early exit)
// IO: 0 = byte planes (uint8 [B][6][N][N]); 2 = TRACKED boards (uint32 [B][5N+1]: the rows of black, white, invalid,
// multi_black, multi_white + the flag word, gg_v4.h) - a lane reads and writes its own five row words, the classes travel with
// the board: no LDS, no first analysis, so even a one-ply launch is just the ply
// SHORT: the launch is one or two plies long (tracked boards only) - a latency chain that lasts as long as its slowest wave, and
// with so few plies SOME wave of the launch takes the rare second flood round of the three-flood ply, so a short launch runs the
// five-flood ply and mixes its draws lane by lane; as an instantiation of its own (not a branch) because a one-ply launch
// also pays for every instruction-cache line it has to jump to (3.19 us per hipGraph node at config 2's size against 3.36)
// WPB: waves per workgroup, each wave an independent group of boards (a launch of single-wave workgroups enters the machine over
// ~0.26 ns per workgroup - tools/exp/oneply_ramp.py - which a ONE-ply launch of a thousand workgroups feels: k_rollout_lat_w4)
template <int R, bool FULLN, bool AUTO, int IO, int WPB, bool SHORT>
__device__ __forceinline__ void rollout_lat_body(uint8_t *__restrict__ states, uint64_t *__restrict__ rng,
                                                 int32_t *__restrict__ last_actions,
                                                 int64_t *__restrict__ steps_done, int64_t B, int N, int plies,
                                                 int auto_reset) {
  using L = Lat<R>;
  constexpr int LPB = L::LPB, NBW = L::NBW, FW = L::FW, NF = L::NF, NREG = L::NREG;
  if (FULLN) N = R;
  if (AUTO) auto_reset = 1;
  __shared__ __attribute__((aligned(16))) uint32_t lds_[WPB][LdsLat<R>::kTotal];
  __shared__ uint2 lut_[WPB][256];
  const int wv = WPB > 1 ? (int)(threadIdx.x >> 6) : 0;
  uint32_t *lds = lds_[wv];
  uint2 *lut = lut_[wv];
  const int lane = threadIdx.x & (kWave - 1);
  const int r = lane & (LPB - 1), j = lane / LPB;      // row, board of the wave
  const int P = N * N, S = 6 * P;
  const uint32_t full = r < N ? (1u << N) - 1u : 0u;
  static_assert(IO == 0 || IO == 2, "byte planes or tracked boards");
  constexpr bool TRACKED = IO == 2;
  const int W = 5 * N + 1;
  GG_PROF_DECL;
  bool tables = TRACKED;
  const int64_t ngroups = (B + NBW - 1) / NBW;
  for (int64_t g = (int64_t)blockIdx.x * WPB + wv; g < ngroups; g += (int64_t)gridDim.x * WPB) {
    const int64_t b_first = g * NBW;
    const bool on = b_first + j < B;
    const int64_t b = on ? b_first + j : B - 1;
    uint8_t *gs = states + b * (int64_t)S;
    // ---------------------------------------------------------------- load: all boards of the wave staged, then one row per lane
    uint32_t *gp = reinterpret_cast<uint32_t *>(states) + b * (int64_t)W;   // TRACKED: this lane's board
    uint64_t x;
    uint32_t me, op, M, inv, fl;
    if (TRACKED) {
      const int rc = r < N ? r : 0;
      x = rng[b];   // (first: every read of the prologue is in flight before the first one is consumed)
      uint32_t bl = gp[rc], wh = gp[N + rc];
      inv = gp[2 * N + rc];
      M = gp[3 * N + rc] | gp[4 * N + rc];
      fl = gp[5 * N] & 7u;
      if (!on || r >= N) { bl = wh = inv = M = 0; }
      if (!on) fl = 0;
      me = (fl & 1u) ? wh : bl;
      op = (fl & 1u) ? bl : wh;
    } else {
      // the boards of the wave are ONE contiguous slice of HBM: its aligned 16-byte vectors, the class masks of the lane's row
      // and the generator are read together (one round trip; each helper waiting for its own reads cost a one-ply launch
      // 3 500 of its 11 300 cycles per wave: tools/exp/oneply_where.py), the spread table is built while they are in flight
      constexpr int NVL = (((15 + NBW * 6 * R * R + 15) >> 4) + kWave - 1) / kWave;
      typedef uint32_t Vec __attribute__((ext_vector_type(4)));
      const uint8_t *g0 = states + b_first * (int64_t)S;
      const int64_t nb = B - b_first < NBW ? B - b_first : NBW;
      const uint32_t mis = (uint32_t)((uintptr_t)g0 & 15u);
      const int nv = (int)((mis + (uint32_t)(nb * S) + 15u) >> 4);
      uint32_t cwm[kCwClasses];
      int rr = r < 19 ? r : 19;
      asm volatile("" : "+v"(rr));   // (loop-variant for the compiler: hoisted out of the loop, the reads would be waited for before the loop)
#pragma unroll
      for (int i = 0; i < kCwClasses; ++i) cwm[i] = kCw.m[i][rr];
      Vec sv[NVL];
#pragma unroll
      for (int k = 0; k < NVL; ++k) {
        const int v = lane + kWave * k;
        sv[k] = *reinterpret_cast<const Vec *>(g0 - mis + 16 * (v < nv ? v : nv - 1));
      }
      x = rng[b];
      if (!tables) { load_spread_lut(lut, lane); tables = true; }
      WAVE_SYNC();
      uint8_t *iob = reinterpret_cast<uint8_t *>(lds + LdsLat<R>::kIo);
#pragma unroll
      for (int k = 0; k < NVL; ++k) {
        const int v = lane + kWave * k;
        if (v < nv) *reinterpret_cast<Vec *>(iob + 16 * v) = sv[k];
      }
      WAVE_SYNC();
      const uint8_t *io = iob + mis + j * S;
      uint32_t bl = plane_to_row<R>(io, N, r), wh = plane_to_row<R>(io + P, N, r);
      inv = plane_to_row<R>(io + 3 * P, N, r);
      fl = (io[2 * P] ? 1u : 0u) | (io[4 * P] ? 2u : 0u) | (io[5 * P] ? 4u : 0u);   // turn, passed, done
      if (!on) { bl = wh = inv = 0; fl = 0; }
      GG_PROF(2);   // bytes -> rows
      M = lat_classes<R>(bl, wh, full, cwm);
      me = (fl & 1u) ? wh : bl;
      op = (fl & 1u) ? bl : wh;
    }
    int lastv = -1, played = 0;
    const int rN = r * N;
    // The draws of a board, LPB plies at a time: the generator is a counter (x += c per draw) and a board draws once per ply
    // from ply 0 until it freezes for good, so the draw of ply t is mix(x0 + (t + 1) c) - lane r of the board computes the one
    // of ply t0 + r every LPB plies, a ply fetches its own with one ds_bpermute (in flight under the popcounts and board scans),
    // and the generator the launch leaves behind is x0 + played c.  (19 VALU instructions per ply -> 2.)
    const uint64_t x0 = x;
    uint32_t uq = 0;
    const int lane_b0 = lane & ~(LPB - 1);
    GG_PROF(6);   // load + first classes
    // ---------------------------------------------------------------- plies
#pragma unroll 1
    for (int t = 0; t < plies; ++t) {
      const bool live = AUTO ? on : (on && !((fl & 4u) && !auto_reset));
      if (!AUTO && __ballot(live) == 0) break;
      if (auto_reset && __ballot(live && (fl & 4u))) {     // auto-reset of a finished game (rare)
        const uint32_t keep = (live && (fl & 4u)) ? 0u : ~0u;
        me &= keep; op &= keep; M &= keep; inv &= keep; fl &= keep;
      }
      const uint32_t lv = live ? ~0u : 0u;
      // 1. the draw: uniform over the valid points + the pass (GoEnv.uniform_random_action; oracle/gg_oracle.c rollout_ply)
      uint32_t uh;
      if (SHORT) {   // (a one- or two-ply launch is a latency chain: every lane mixes the ply's draw itself, no hand-off)
        uint64_t xx = x0 + (uint64_t)(uint32_t)t * 0x9E3779B97F4A7C15ull;
        uh = (uint32_t)(splitmix_next(xx) >> 32);
      } else {
        if ((t & (LPB - 1)) == 0) {
          uint64_t xx = x0 + (uint64_t)(uint32_t)(t + r) * 0x9E3779B97F4A7C15ull;   // (splitmix_next adds the (t + r + 1)-th c itself)
          uq = (uint32_t)(splitmix_next(xx) >> 32);
        }
        uh = (uint32_t)__builtin_amdgcn_ds_bpermute((lane_b0 + (t & (LPB - 1))) << 2, (int)uq);
      }
      const uint32_t valid = full & ~inv;
      const uint32_t cnt = (uint32_t)__popc(valid);
      const uint32_t incl = lat_board_scan<LPB>(cnt);
      const uint32_t total = lat_board_sum<LPB>(cnt);
      const uint32_t k = __umulhi(uh, total + 1u);
      const uint32_t tt = k - (incl - cnt);
      const bool hit = live && tt < cnt;                 // this lane's row holds the k-th valid point
      const uint32_t pos = lat_kth_bit<L::kBits>(valid, tt);
      const uint32_t Q = hit ? (1u << pos) : 0u;
      const bool pass = k == total;                      // (the same in every lane of the board)
      {   // the action of the board's last live ply: the hit lane holds it, every lane holds a pass, the others -1
        const int cand = hit ? rN + (int)pos : (pass ? P : -1);
        lastv = (int)B3(lv, (uint32_t)cand, (uint32_t)lastv, T_SEL);
      }
      played -= (int)lv;
      GG_PROF(0);
      if (SHORT) lat_play_full<R>(me, op, M, inv, fl, Q, pass, lv, full);
      else lat_play<R>(me, op, M, inv, fl, Q, pass, lv, full);
      GG_PROF(4);
    }
    GG_PROF(5);
    // ---------------------------------------------------------------- store
    {
      const uint32_t turn = fl & 1u;
      const uint32_t bl = turn ? op : me, wh = turn ? me : op;
      const int lastb = lat_board_max<LPB>(lastv);
      if (TRACKED) {
        if (on && played != 0 && r < N) {
          gp[r] = bl; gp[N + r] = wh; gp[2 * N + r] = inv;
          gp[3 * N + r] = M & bl; gp[4 * N + r] = M & wh;
        }
        if (on && played != 0 && r == 0) gp[5 * N] = fl;
      } else if (__ballot(played != 0)) {
        lat_emit<R>(gs, bl, wh, inv, turn, (fl >> 1) & 1u, (fl >> 2) & 1u, full, N, r,
                    lds + LdsLat<R>::kBs + j * L::kBsWords, lut, on && played != 0);
      }
      if (on && r == 0) {
        rng[b] = x0 + (uint64_t)(uint32_t)played * 0x9E3779B97F4A7C15ull;
        if (last_actions) last_actions[b] = lastb;
        if (steps_done && played) atomicAdd(reinterpret_cast<unsigned long long *>(steps_done) + b, (unsigned long long)played);
      }
    }
    GG_PROF(7);   // write-back
  }
  GG_PROF_FLUSH;
}
template <int R, bool FULLN, bool AUTO, int IO = 0, bool SHORT = false>
__global__ __launch_bounds__(kWave, 4) void k_rollout_lat(uint8_t *__restrict__ states, uint64_t *__restrict__ rng,
                                                          int32_t *__restrict__ last_actions,
                                                          int64_t *__restrict__ steps_done, int64_t B, int N, int plies,
                                                          int auto_reset) {
  rollout_lat_body<R, FULLN, AUTO, IO, 1, SHORT>(states, rng, last_actions, steps_done, B, N, plies, auto_reset);
}
// tracked boards, a few plies per launch: four waves per workgroup
template <int R, bool FULLN, bool AUTO, bool SHORT>
__global__ __launch_bounds__(4 * kWave, 4) void k_rollout_lat_w4(uint8_t *__restrict__ states, uint64_t *__restrict__ rng,
                                                                 int32_t *__restrict__ last_actions,
                                                                 int64_t *__restrict__ steps_done, int64_t B, int N, int plies,
                                                                 int auto_reset) {
  rollout_lat_body<R, FULLN, AUTO, 2, 4, SHORT>(states, rng, last_actions, steps_done, B, N, plies, auto_reset);
}

// Tromp-Taylor areas (gym_go/gogame.py:275-300) of the wave's boards in this layout: a colour owns its stones plus the empty
// regions that touch only that colour, and a region touches a colour iff the flood of the EMPTY points seeded next to that
// colour's stones covers it - two floods in two fields of one register (19x19: two registers), summed per board.
template <int R>
__device__ __forceinline__ void lat_areas(uint32_t bl, uint32_t wh, uint32_t full, uint32_t &area_b, uint32_t &area_w) {
  using L = Lat<R>;
  constexpr int LPB = L::LPB, FW = L::FW, K = L::NF >= 2 ? 1 : 2;
  const uint32_t E = full & ~(bl | wh);
  const uint32_t sb = lat_dilate<LPB>(bl) & E, sw = lat_dilate<LPB>(wh) & E;
  uint32_t F[K], Mk[K], Mkr[K];
  if (K == 1) {
    F[0] = sb | (sw << (FW & 31));
    Mk[0] = E | (E << (FW & 31));
  } else {
    F[0] = sb; F[K - 1] = sw;
    Mk[0] = E; Mk[K - 1] = E;
  }
#pragma unroll
  for (int k = 0; k < K; ++k) Mkr[k] = __brev(Mk[k]);
  lat_flood<LPB, K>(F, Mk, Mkr);
  const uint32_t fb = K == 1 ? (F[0] & L::FM) : F[0], fw = K == 1 ? (F[0] >> (FW & 31)) : F[K - 1];
  const uint32_t cb = (uint32_t)__popc(bl) + (uint32_t)__popc(fb & ~fw), cw = (uint32_t)__popc(wh) + (uint32_t)__popc(fw & ~fb);
  const uint32_t sum = lat_board_sum<LPB>(cb | (cw << 16));     // (<= 361 each)
  area_b = sum & 0xFFFFu;
  area_w = sum >> 16;
}

// GoEnv.step for every game of a batched env whose boards are kept TRACKED (gg_batch_env_step_tracked; gym_go/envs/go_env.py:49-76)
// on batches that leave the SIMDs under-filled: ONE ply of the kernel above - the move given (MOVES: env.actions) or drawn -
// with GoEnv.step's outputs: status, dones, the action used, GoEnv.reward (:128-149; Tromp-Taylor areas of the resulting
// position when the game has ended or the reward is `heuristic`) and, when asked, the byte-plane observation of every game.
// Semantics as k_rollout4<R, 2, MOVES, FULLN, true> (gg_v4.h): a finished game is reset first when auto_reset - and the reset
// stands even when the given move is then refused (GoEnv.reset comes before the action check) - or refuses the step.
template <int R, bool FULLN, bool MOVES, int WPB>
__device__ __forceinline__ void env_step_lat_body(uint32_t *__restrict__ tracked, uint64_t *__restrict__ rng,
                                                  int64_t *__restrict__ steps_done, int64_t B, int N, int auto_reset,
                                                  const EnvArgs &env) {
  using L = Lat<R>;
  constexpr int LPB = L::LPB, NBW = L::NBW;
  if (FULLN) N = R;
  __shared__ __attribute__((aligned(16))) uint32_t bsv_[WPB][NBW * L::kBsWords];
  __shared__ uint2 lut_[WPB][256];
  const int wv = WPB > 1 ? (int)(threadIdx.x >> 6) : 0;
  uint32_t *bsv = bsv_[wv];
  uint2 *lut = lut_[wv];
  const int lane = threadIdx.x & (kWave - 1);
  const int r = lane & (LPB - 1), j = lane / LPB;
  const int P = N * N, S = 6 * P, W = 5 * N + 1;
  const uint32_t full = r < N ? (1u << N) - 1u : 0u;
  const bool obs = env.states_out != nullptr;
  if (obs) load_spread_lut(lut, lane);
  const int64_t ngroups = (B + NBW - 1) / NBW;
  for (int64_t g = (int64_t)blockIdx.x * WPB + wv; g < ngroups; g += (int64_t)gridDim.x * WPB) {
    const int64_t b_first = g * NBW;
    const bool on = b_first + j < B;
    const int64_t b = on ? b_first + j : B - 1;
    uint32_t *gp = tracked + b * (int64_t)W;
    // (the generator / the given move first: every read of the prologue is in flight before the first one is consumed)
    uint64_t x = MOVES ? 0 : rng[b];
    const int mv_in = MOVES ? env.actions[b] : 0;
    uint32_t me, op, M, inv, fl;
    {
      const int rc = r < N ? r : 0;
      uint32_t bl = gp[rc], wh = gp[N + rc];
      inv = gp[2 * N + rc];
      M = gp[3 * N + rc] | gp[4 * N + rc];
      fl = gp[5 * N] & 7u;
      if (!on || r >= N) { bl = wh = inv = M = 0; }
      if (!on) fl = 0;
      me = (fl & 1u) ? wh : bl;
      op = (fl & 1u) ? bl : wh;
    }
    const bool done0 = (fl & 4u) != 0;
    bool live, reset, pass = false;
    uint32_t Q = 0;
    int taken = -1;
    if (MOVES) {
      const int mv = mv_in;
      reset = on && done0 && auto_reset != 0;
      const bool inrange = mv >= 0 && mv <= P;
      const bool point = inrange && mv < P;
      const int ar = point ? mv / N : 0, ac = point ? mv - ar * N : 0;
      // the lane that owns row ar tests the mask bit, the board shares the verdict (a board being reset is empty)
      const uint32_t bad = (point && !reset && r == ar) ? ((inv >> ac) & 1u) : 0u;
      const bool illegal = lat_board_sum<LPB>(bad) != 0u;
      live = on && (!done0 || reset) && inrange && !illegal;
      pass = live && mv == P;
      Q = (live && point && r == ar) ? (1u << ac) : 0u;
      taken = mv;
    } else {
      live = on && !(done0 && !auto_reset);
      reset = live && done0;
    }
    if (__ballot(reset)) {   // GoEnv.reset: the board is init_state from now on (also when the move is then refused)
      const uint32_t keep = reset ? 0u : ~0u;
      me &= keep; op &= keep; M &= keep; inv &= keep; fl &= keep;
    }
    const uint32_t lv = live ? ~0u : 0u;
    if (!MOVES) {   // the draw of k_rollout_lat
      const uint32_t valid = full & ~inv;
      const uint32_t cnt = (uint32_t)__popc(valid);
      const uint32_t incl = lat_board_scan<LPB>(cnt);
      const uint32_t total = lat_board_sum<LPB>(cnt);
      uint64_t xn = x;
      const uint64_t u = splitmix_next(xn);
      if (live) x = xn;
      const uint32_t k = __umulhi((uint32_t)(u >> 32), total + 1u);
      const uint32_t tt = k - (incl - cnt);
      const bool hit = live && tt < cnt;
      const uint32_t pos = lat_kth_bit<L::kBits>(valid, tt);
      Q = hit ? (1u << pos) : 0u;
      pass = k == total;
      taken = lat_board_max<LPB>(!live ? -1 : (hit ? r * N + (int)pos : (pass ? P : -1)));
    }
    if (__ballot(live)) lat_play_full<R>(me, op, M, inv, fl, Q, pass, lv, full);   // (one ply per launch: see rollout_lat_body)
    // ---- the boards after the step, GoEnv.step's outputs
    const uint32_t turn = fl & 1u;
    const uint32_t bl = turn ? op : me, wh = turn ? me : op;
    const bool doneb = (fl & 4u) != 0;
    uint32_t area_b = 0, area_w = 0;
    if (__ballot(on && (env.heuristic != 0 || doneb))) lat_areas<R>(bl, wh, full, area_b, area_w);
    if (on && r == 0) {
      const float margin = (float)((int)area_b - (int)area_w) - env.komi;
      float rwd;   // GoEnv.reward (gym_go/envs/go_env.py:128-149), black's perspective
      if (env.heuristic) rwd = doneb ? (margin > 0.f ? 1.f : -1.f) * (float)P : margin;
      else rwd = doneb ? (margin > 0.f ? 1.f : (margin < 0.f ? -1.f : 0.f)) : 0.f;
      if (env.rewards) env.rewards[b] = rwd;
      if (env.dones) env.dones[b] = (uint8_t)doneb;
      if (env.status) env.status[b] = live ? GG_STATUS_OK : GG_STATUS_ILLEGAL;
      if (env.taken) env.taken[b] = taken;
      if (!MOVES && live) rng[b] = x;
      if (steps_done && live) atomicAdd(reinterpret_cast<unsigned long long *>(steps_done) + b, 1ull);
      if (live || reset) gp[5 * N] = fl;
    }
    if (on && (live || reset) && r < N) {
      gp[r] = bl; gp[N + r] = wh; gp[2 * N + r] = inv;
      gp[3 * N + r] = M & bl; gp[4 * N + r] = M & wh;
    }
    if (obs)   // the observation: every board of the wave as byte planes
      lat_emit<R>(env.states_out + b * (int64_t)S, bl, wh, inv, turn, (fl >> 1) & 1u, (fl >> 2) & 1u, full, N, r,
                  bsv + j * L::kBsWords, lut, on);
  }
}
template <int R, bool FULLN, bool MOVES>
__global__ __launch_bounds__(kWave, 4) void k_env_step_lat(uint32_t *__restrict__ tracked, uint64_t *__restrict__ rng,
                                                           int64_t *__restrict__ steps_done, int64_t B, int N, int auto_reset,
                                                           EnvArgs env) {
  env_step_lat_body<R, FULLN, MOVES, 1>(tracked, rng, steps_done, B, N, auto_reset, env);
}
// ... four waves per workgroup (small batches: the dispatcher's ramp, rollout_lat_body)
template <int R, bool FULLN, bool MOVES>
__global__ __launch_bounds__(4 * kWave, 4) void k_env_step_lat_w4(uint32_t *__restrict__ tracked, uint64_t *__restrict__ rng,
                                                                  int64_t *__restrict__ steps_done, int64_t B, int N, int auto_reset,
                                                                  EnvArgs env) {
  env_step_lat_body<R, FULLN, MOVES, 4>(tracked, rng, steps_done, B, N, auto_reset, env);
}

}  // namespace gg
