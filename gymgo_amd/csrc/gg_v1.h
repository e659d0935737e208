// gg_v1.h - kernel family v1: ONE WAVEFRONT PER BOARD.
//
// 40 flood lanes per board: lanes 0-19 flood the next mover's groups, lanes 32-51 the mover's groups, one lane per
// liberty class (bit k of the row / column index == v, k < 5, v in {0,1}).  A group reached from both classes
// (k,0) and (k,1) for some k has >= 2 distinct liberties; a group reached by neither class of k = 0 has none
// (captured); everything else has exactly one - all the reference's invalid-move rule needs
// (gym_go/state_utils.py:24-83 restated point-wise).  The single-board kernels (areas, invalid mask,
// sampler, stand-alone capture resolution) live here too; GG_KERNEL_VARIANT=1 selects this family for the step /
// rollout / children entry points (default is v2, gg_v2.h).
#pragma once
#include "gg_common.h"

namespace gg {

constexpr int kClassBits = 5;                 // row / column indices < 32
constexpr int kClasses = 4 * kClassBits;      // (row|col bit k) x (value v) = 20 floods per colour

struct LaneClass {
  uint32_t rowsel;   // bit r set: row r belongs to this lane's liberty class
  uint32_t colmask;  // columns of this lane's liberty class
  bool second;       // lanes 32..63 flood the second colour
};

__device__ __forceinline__ LaneClass make_lane_class(int lane) {
  const uint32_t pat[kClassBits] = {0xAAAAAAAAu, 0xCCCCCCCCu, 0xF0F0F0F0u, 0xFF00FF00u, 0xFFFF0000u};
  LaneClass lc;
  int cls = lane & 31, k = cls >> 1, v = cls & 1;
  lc.second = lane >= 32;
  lc.rowsel = 0;
  lc.colmask = 0;
  if (cls < kClasses) {
    uint32_t p = 0;
#pragma unroll
    for (int i = 0; i < kClassBits; ++i)
      if ((k % kClassBits) == i) p = pat[i];
    p = v ? p : ~p;
    if (k < kClassBits) { lc.rowsel = p; lc.colmask = 0xFFFFFFFFu; }
    else { lc.rowsel = 0xFFFFFFFFu; lc.colmask = p; }
  }
  return lc;
}

// Liberty analysis of the whole board (L1 in, L1 out).  c0 / c1 = stones of the two colours
// (lane r = row r), e = empty points.  Returns for lane r < R:
//   multi0 / multi1: stones of c0 / c1 whose group has >= 2 distinct liberties
//   alive0:          stones of c0 whose group has >= 1 liberty
// L1 -> L2 goes through LDS: the rows (and their bit-reversals) are written once and every flood
// lane fetches all rows of its colour with broadcast 16-byte reads.
template <int R>
__device__ __forceinline__ void analyze(uint32_t c0, uint32_t c1, uint32_t e, const LaneClass lc, uint32_t *sc,
                                        uint32_t *rows5, int lane, uint32_t &multi0, uint32_t &alive0,
                                        uint32_t &multi1) {
  constexpr int RS = Cfg<R>::kRowStride;
  constexpr int RV = (R + 3) / 4;  // uint4 reads per plane
  WAVE_SYNC();                     // earlier readers of rows5 / sc are done
  if (lane < 32) {
    rows5[lane] = c0;
    rows5[32 + lane] = c1;
    rows5[64 + lane] = __brev(c0);
    rows5[96 + lane] = __brev(c1);
    rows5[128 + lane] = e;
  }
  WAVE_SYNC();
  uint32_t m[RV * 4], mrev[RV * 4], ee[RV * 4];
  {
    const uint4 *pm = reinterpret_cast<const uint4 *>(rows5 + (lc.second ? 32 : 0));
    const uint4 *pr = reinterpret_cast<const uint4 *>(rows5 + 64 + (lc.second ? 32 : 0));
    const uint4 *pe = reinterpret_cast<const uint4 *>(rows5 + 128);
#pragma unroll
    for (int i = 0; i < RV; ++i) {
      uint4 a = pm[i], b = pr[i], c = pe[i];
      m[4 * i] = a.x; m[4 * i + 1] = a.y; m[4 * i + 2] = a.z; m[4 * i + 3] = a.w;
      mrev[4 * i] = b.x; mrev[4 * i + 1] = b.y; mrev[4 * i + 2] = b.z; mrev[4 * i + 3] = b.w;
      ee[4 * i] = c.x; ee[4 * i + 1] = c.y; ee[4 * i + 2] = c.z; ee[4 * i + 3] = c.w;
    }
  }
  uint32_t mm[R], mr[R], f[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    mm[r] = m[r];
    mr[r] = mrev[r];
    uint32_t rowon = 0u - ((lc.rowsel >> r) & 1u);
    ee[r] &= rowon & lc.colmask;  // empties of this lane's liberty class
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    uint32_t x = (ee[r] << 1) | (r > 0 ? ee[r - 1] : 0u);
    uint32_t y = (ee[r] >> 1) | (r < R - 1 ? ee[r + 1] : 0u);
    f[r] = mm[r] & (x | y);  // stones touching a liberty of the class
  }
  flood2_dual<R>(mm, mr, f, sc + lane * RS);
  WAVE_SYNC();
  multi0 = 0; multi1 = 0; alive0 = 0;
  if (lane < R) {
#pragma unroll
    for (int k = 0; k < kClasses / 2; ++k) {
      uint32_t a = sc[(2 * k) * RS + lane], b = sc[(2 * k + 1) * RS + lane];
      multi0 |= a & b;
      if (k == 0) alive0 = a | b;
      uint32_t a1 = sc[(32 + 2 * k) * RS + lane], b1 = sc[(32 + 2 * k + 1) * RS + lane];
      multi1 |= a1 & b1;
    }
  }
}

struct Geo {
  int N, P;
  uint32_t inv;      // ceil(2^16 / N): row = (a * inv) >> 16 for every a <= N*N (checked on the host)
  uint32_t full_l1;  // lane < N ? (1<<N)-1 : 0
};

// compute_invalid_moves (gym_go/state_utils.py:24-83) in closed form for the side `nx` that moves next:
// invalid = occupied or (no neighbour is: empty | nx stone with >=2 liberties | other stone with ==1).
__device__ __forceinline__ uint32_t invalid_from(uint32_t nx, uint32_t pl, uint32_t multi_nx, uint32_t multi_pl,
                                                 const Geo &g, int lane) {
  uint32_t e = g.full_l1 & ~(nx | pl);
  uint32_t x = e | (nx & multi_nx) | (pl & ~multi_pl);
  uint32_t up = __shfl_up(x, 1), dn = __shfl_down(x, 1);
  if (lane == 0) up = 0;
  if (lane >= g.N - 1) dn = 0;
  uint32_t nb = (x << 1) | (x >> 1) | up | dn;
  return g.full_l1 & ~(e & nb);
}

// One transition on L1 bitboards (gym_go/gogame.py:34-87 without the plane bookkeeping).
// mine = mover's stones, opp = next mover's stones; both updated.  Returns the invalid mask for the
// next mover (incl. ko).  `a` must be a legal point or P (pass).
template <int R>
__device__ __forceinline__ uint32_t step_core(uint32_t &mine, uint32_t &opp, int a, const Geo &g, const LaneClass lc,
                                              uint32_t *sc, uint32_t *rows5, int lane) {
  const bool is_pass = a == g.P;
  int ko_r = -1, ko_c = 0;
  bool boxed = false;
  if (!is_pass) {
    int ra = (int)(((uint32_t)a * g.inv) >> 16), ca = a - ra * g.N;
    uint32_t bit = 1u << ca;
    if (lane == ra) mine |= bit;                       // gogame.py:62
    // state_utils.adj_data :214-223 - every on-board neighbour holds an opponent stone
    uint32_t nbm = 0;
    if (lane == ra) nbm = (bit << 1) | (bit >> 1);
    if (lane == ra - 1 || lane == ra + 1) nbm = bit;
    nbm &= g.full_l1;
    boxed = __ballot((nbm & ~opp) != 0) == 0;
  }
  uint32_t multi_opp, alive_opp, multi_mine;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    uint32_t e = g.full_l1 & ~(mine | opp);
    analyze<R>(opp, mine, e, lc, sc, rows5, lane, multi_opp, alive_opp, multi_mine);
    if (pass == 0 && !is_pass) {
      // state_utils.update_pieces :159-180 - opponent groups left without a liberty die.  Only groups
      // touching the new stone can be in that state (every group had a liberty before the move).
      uint32_t dead = opp & ~alive_opp;
      uint64_t dm = __ballot(dead != 0);
      if (dm) {
        // gogame.py:72-75 - ko iff exactly one stone died and the new stone is boxed in
        uint64_t multi_rows = __ballot(__popc(dead) > 1);
        if (boxed && multi_rows == 0 && (dm & (dm - 1)) == 0) {
          ko_r = __ffsll((unsigned long long)dm) - 1;
          uint32_t drow = __builtin_amdgcn_readlane(dead, ko_r);
          ko_c = __ffs(drow) - 1;
        }
        opp &= ~dead;
        continue;  // liberties changed: analyse the board again
      }
    }
    break;
  }
  uint32_t invalid = invalid_from(opp, mine, multi_opp, multi_mine, g, lane);
  if (lane == ko_r) invalid |= 1u << ko_c;  // state_utils.py:81-82
  return invalid;
}

// ---------------------------------------------------------------- kernels

struct PlaneBytes { uint8_t turn, passed, done; };

// Emit a whole 6-plane board into the LDS staging buffer (ob[j] = board byte j) from L1 rows.
template <int R>
__device__ __forceinline__ void emit_board(uint8_t *ob, uint32_t black, uint32_t white, uint32_t invalid,
                                           PlaneBytes pb, const Geo &g, int lane) {
  WAVE_SYNC();
  row_to_plane<R>(ob, black, g.N, lane);
  row_to_plane<R>(ob + g.P, white, g.N, lane);
  splat_plane(ob + 2 * g.P, pb.turn, g.P, lane);
  row_to_plane<R>(ob + 3 * g.P, invalid, g.N, lane);
  splat_plane(ob + 4 * g.P, pb.passed, g.P, lane);
  splat_plane(ob + 5 * g.P, pb.done, g.P, lane);
  WAVE_SYNC();
}

// Read the uniform-plane flags + the INVD byte of point `pt` of a board in HBM:
// bit0 turn, bit1 INVD[pt], bit2 previous move was a pass, bit3 game over.
__device__ __forceinline__ uint32_t load_flags(const uint8_t *g, int P, int pt, int lane) {
  uint8_t fb = 0;
  if (lane < 4) {
    int off = lane == 0 ? 2 * P : lane == 1 ? 3 * P + pt : lane == 2 ? 4 * P : 5 * P;
    fb = g[off];
  }
  return (uint32_t)__ballot(fb != 0) & 0xFu;
}

template <int R>
__global__ __launch_bounds__(kWave) void k_next_states(const uint8_t *__restrict__ in,
                                                       const int32_t *__restrict__ actions,
                                                       uint8_t *__restrict__ out, int32_t *__restrict__ status,
                                                       int64_t B, int N, uint32_t inv, int canonical) {
  __shared__ __attribute__((aligned(16))) uint8_t iobuf[Cfg<R>::kIoBytes];
  __shared__ __attribute__((aligned(16))) uint32_t sc[kWave * Cfg<R>::kRowStride];
  __shared__ __attribute__((aligned(16))) uint32_t rows5[160];
  const int lane = threadIdx.x;
  Geo g;
  g.N = N; g.P = N * N; g.inv = inv;
  g.full_l1 = lane < N ? (1u << N) - 1u : 0u;
  const int S = 6 * g.P;
  const LaneClass lc = make_lane_class(lane);

  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    const uint8_t *gi = in + b * (int64_t)S;
    uint8_t *go = out + b * (int64_t)S;
    int a = __builtin_amdgcn_readfirstlane(actions[b]);
    const bool in_range = a >= 0 && a <= g.P;
    const bool is_pass = a == g.P;
    uint32_t flags = load_flags(gi, g.P, (in_range && !is_pass) ? a : 0, lane);
    if (!in_range || (!is_pass && (flags & 2u))) {
      // gogame.py:59 / :117 would raise: row passes through unchanged, status flags it
      for (int i = lane; i < S; i += kWave) go[i] = gi[i];
      if (status && lane == 0) status[b] = GG_STATUS_ILLEGAL;
      continue;
    }
    WAVE_SYNC();
    const uint32_t mi = stage_in(gi, 2 * g.P, iobuf, lane);
    WAVE_SYNC();
    uint32_t black = plane_to_row<R>(iobuf + mi, N, lane);
    uint32_t white = plane_to_row<R>(iobuf + mi + g.P, N, lane);
    const int pl = flags & 1u;                       // gogame.py:44 turn
    uint32_t mine = pl ? white : black, opp = pl ? black : white;
    uint32_t invalid = step_core<R>(mine, opp, a, g, lc, sc, rows5, lane);
    black = pl ? opp : mine;
    white = pl ? mine : opp;
    PlaneBytes pb;
    pb.passed = is_pass ? 1 : 0;                                            // gogame.py:50 / :56
    pb.done = ((flags & 8u) || (is_pass && (flags & 4u))) ? 1 : 0;          // gogame.py:51-53 (sticky)
    int nturn = 1 - pl;                                                     // state_utils.py:235-241
    if (canonical && nturn == 1) {                                          // gogame.py:313-321
      uint32_t t = black; black = white; white = t;
      nturn = 0;
    }
    pb.turn = (uint8_t)nturn;
    emit_board<R>(iobuf + ((uintptr_t)go & 15u), black, white, invalid, pb, g, lane);
    stage_out(go, S, iobuf, lane);
    if (status && lane == 0) status[b] = GG_STATUS_OK;
  }
}

template <int R>
__global__ __launch_bounds__(kWave) void k_invalid_mask(const uint8_t *__restrict__ states,
                                                        const int32_t *__restrict__ ko, uint8_t *__restrict__ mask,
                                                        int64_t B, int N, uint32_t inv) {
  __shared__ __attribute__((aligned(16))) uint8_t iobuf[Cfg<R>::kIoBytes];
  __shared__ __attribute__((aligned(16))) uint32_t sc[kWave * Cfg<R>::kRowStride];
  __shared__ __attribute__((aligned(16))) uint32_t rows5[160];
  const int lane = threadIdx.x;
  Geo g;
  g.N = N; g.P = N * N; g.inv = inv;
  g.full_l1 = lane < N ? (1u << N) - 1u : 0u;
  const int S = 6 * g.P;
  const LaneClass lc = make_lane_class(lane);
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    const uint8_t *gi = states + b * (int64_t)S;
    uint32_t flags = load_flags(gi, g.P, 0, lane);
    WAVE_SYNC();
    const uint32_t mi = stage_in(gi, 2 * g.P, iobuf, lane);
    WAVE_SYNC();
    uint32_t black = plane_to_row<R>(iobuf + mi, N, lane);
    uint32_t white = plane_to_row<R>(iobuf + mi + g.P, N, lane);
    const int nx = flags & 1u;  // side to move
    uint32_t nxs = nx ? white : black, pls = nx ? black : white;
    uint32_t e = g.full_l1 & ~(black | white);
    uint32_t multi_nx, alive_nx, multi_pl;
    analyze<R>(nxs, pls, e, lc, sc, rows5, lane, multi_nx, alive_nx, multi_pl);
    uint32_t invalid = invalid_from(nxs, pls, multi_nx, multi_pl, g, lane);
    if (ko) {
      int k = __builtin_amdgcn_readfirstlane(ko[b]);
      if (k >= 0 && k < g.P) {
        int kr, kc;
        split_action(k, N, inv, kr, kc);
        if (lane == kr) invalid |= 1u << kc;
      }
    }
    uint8_t *gm = mask + b * (int64_t)g.P;
    WAVE_SYNC();
    row_to_plane<R>(iobuf + ((uintptr_t)gm & 15u), invalid, N, lane);
    WAVE_SYNC();
    stage_out(gm, g.P, iobuf, lane);
  }
}

// gogame.areas (gym_go/gogame.py:275-300): flood the empty points from those touching black (lane 0)
// and those touching white (lane 1); a region reached by exactly one colour belongs to it.
template <int R>
__global__ __launch_bounds__(kWave) void k_areas(const uint8_t *__restrict__ states, int32_t *__restrict__ black_area,
                                                 int32_t *__restrict__ white_area, int64_t B, int N) {
  __shared__ __attribute__((aligned(16))) uint8_t iobuf[Cfg<R>::kIoBytes];
  __shared__ __attribute__((aligned(16))) uint32_t sc[kWave * Cfg<R>::kRowStride];
  const int lane = threadIdx.x;
  Geo g;
  g.N = N; g.P = N * N; g.inv = 0;
  g.full_l1 = lane < N ? (1u << N) - 1u : 0u;
  const int S = 6 * g.P;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    const uint8_t *gi = states + b * (int64_t)S;
    WAVE_SYNC();
    const uint32_t mi = stage_in(gi, 2 * g.P, iobuf, lane);
    WAVE_SYNC();
    uint32_t black = plane_to_row<R>(iobuf + mi, N, lane);
    uint32_t white = plane_to_row<R>(iobuf + mi + g.P, N, lane);
    uint32_t e = g.full_l1 & ~(black | white);
    uint32_t m[R], mrev[R], f[R], src[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      m[r] = __builtin_amdgcn_readlane(e, r);
      mrev[r] = __brev(m[r]);
      uint32_t sb = __builtin_amdgcn_readlane(black, r), sw = __builtin_amdgcn_readlane(white, r);
      src[r] = lane == 0 ? sb : lane == 1 ? sw : 0u;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      uint32_t nb = (src[r] << 1) | (src[r] >> 1);
      if (r > 0) nb |= src[r - 1];
      if (r < R - 1) nb |= src[r + 1];
      f[r] = m[r] & nb;
    }
    flood2_dual<R>(m, mrev, f, sc + lane * Cfg<R>::kRowStride);  // f stays valid in registers too
    int ba = 0, wa = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      uint32_t fb = __builtin_amdgcn_readlane(f[r], 0), fw = __builtin_amdgcn_readlane(f[r], 1);
      ba += __popc(fb & ~fw);
      wa += __popc(fw & ~fb);
    }
    // stone counts: sum of per-row popcounts over lanes
    int sb = __popc(black), sw = __popc(white);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
      sb += __shfl_xor(sb, off);
      sw += __shfl_xor(sw, off);
    }
    if (lane == 0) {
      black_area[b] = ba + sb;
      white_area[b] = wa + sw;
    }
  }
}

// gogame.children (gym_go/gogame.py:175-186), padded: work item = (parent, chunk of actions); the parent
// is staged and converted once, each action of the chunk is one step_core on a copy of the bitboards.
template <int R>
__global__ __launch_bounds__(kWave) void k_children(const uint8_t *__restrict__ states, uint8_t *__restrict__ children,
                                                    int64_t B, int N, uint32_t inv, int canonical, int chunks) {
  __shared__ __attribute__((aligned(16))) uint8_t iobuf[Cfg<R>::kIoBytes];
  __shared__ __attribute__((aligned(16))) uint32_t sc[kWave * Cfg<R>::kRowStride];
  __shared__ __attribute__((aligned(16))) uint32_t rows5[160];
  const int lane = threadIdx.x;
  Geo g;
  g.N = N; g.P = N * N; g.inv = inv;
  g.full_l1 = lane < N ? (1u << N) - 1u : 0u;
  const int S = 6 * g.P;
  const int A = g.P + 1;
  const LaneClass lc = make_lane_class(lane);
  const int per = (A + chunks - 1) / chunks;
  for (int64_t w = blockIdx.x; w < B * chunks; w += gridDim.x) {
    const int64_t b = w / chunks;
    const int ch = (int)(w - b * chunks);
    const uint8_t *gi = states + b * (int64_t)S;
    uint32_t flags = load_flags(gi, g.P, 0, lane);
    WAVE_SYNC();
    // planes 0,1 for the stones and plane 3 for slot validity (planes 0..3 are contiguous)
    const uint32_t mi = stage_in(gi, 4 * g.P, iobuf, lane);
    WAVE_SYNC();
    uint32_t black = plane_to_row<R>(iobuf + mi, N, lane);
    uint32_t white = plane_to_row<R>(iobuf + mi + g.P, N, lane);
    uint32_t invd = plane_to_row<R>(iobuf + mi + 3 * g.P, N, lane);
    const int pl = flags & 1u;
    const int a0 = ch * per, a1 = min(A, a0 + per);
    bool zeroed = false;
#pragma unroll 1
    for (int a = a0; a < a1; ++a) {
      uint8_t *go = children + (b * A + a) * (int64_t)S;
      bool valid = true;
      if (a < g.P) {
        int ra, ca;
        split_action(a, N, inv, ra, ca);
        uint32_t row = __builtin_amdgcn_readlane(invd, ra);
        valid = ((row >> ca) & 1u) == 0;
      }
      if (!valid) {
        if (!zeroed) {
          WAVE_SYNC();
          for (int i = lane; i < Cfg<R>::kIoBytes / 16; i += kWave)
            reinterpret_cast<V16a *>(iobuf)[i] = V16a{{0, 0, 0, 0}};
          WAVE_SYNC();
          zeroed = true;
        }
        stage_out(go, S, iobuf, lane);
        continue;
      }
      zeroed = false;
      const bool is_pass = a == g.P;
      uint32_t mine = pl ? white : black, opp = pl ? black : white;
      uint32_t invalid = step_core<R>(mine, opp, a, g, lc, sc, rows5, lane);
      uint32_t nb = pl ? opp : mine, nw = pl ? mine : opp;
      PlaneBytes pb;
      pb.passed = is_pass ? 1 : 0;
      pb.done = ((flags & 8u) || (is_pass && (flags & 4u))) ? 1 : 0;
      int nturn = 1 - pl;
      if (canonical && nturn == 1) {
        uint32_t t = nb; nb = nw; nw = t;
        nturn = 0;
      }
      pb.turn = (uint8_t)nturn;
      emit_board<R>(iobuf + ((uintptr_t)go & 15u), nb, nw, invalid, pb, g, lane);
      stage_out(go, S, iobuf, lane);
    }
  }
}

// state_utils.update_pieces / batch_update_pieces (gym_go/state_utils.py:159-211) as a stand-alone entry: the stone
// of `player` is already on the board at `point`; opponent groups touching it that have no liberty are removed IN
// PLACE (planes 0/1 only) and reported in `killed` (0/1 per point, nullable).  point < 0 or >= N*N: nothing to do.
template <int R>
__global__ __launch_bounds__(kWave) void k_update_pieces(uint8_t *__restrict__ states,
                                                         const int32_t *__restrict__ points,
                                                         const int32_t *__restrict__ players,
                                                         uint8_t *__restrict__ killed, int64_t B, int N, uint32_t inv) {
  __shared__ __attribute__((aligned(16))) uint8_t iobuf[Cfg<R>::kIoBytes];
  __shared__ __attribute__((aligned(16))) uint32_t sc[kWave * Cfg<R>::kRowStride];
  __shared__ __attribute__((aligned(16))) uint32_t rows5[160];
  const int lane = threadIdx.x;
  Geo g;
  g.N = N; g.P = N * N; g.inv = inv;
  g.full_l1 = lane < N ? (1u << N) - 1u : 0u;
  const int S = 6 * g.P;
  const LaneClass lc = make_lane_class(lane);
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    uint8_t *gs = states + b * (int64_t)S;
    const int a = __builtin_amdgcn_readfirstlane(points[b]);
    const int pl = __builtin_amdgcn_readfirstlane(players[b]) & 1;
    WAVE_SYNC();
    const uint32_t mi = stage_in(gs, 2 * g.P, iobuf, lane);
    WAVE_SYNC();
    uint32_t black = plane_to_row<R>(iobuf + mi, N, lane);
    uint32_t white = plane_to_row<R>(iobuf + mi + g.P, N, lane);
    uint32_t mine = pl ? white : black, opp = pl ? black : white;
    uint32_t dead = 0;
    if (a >= 0 && a < g.P) {
      int ra, ca;
      split_action(a, N, inv, ra, ca);
      const uint32_t bit = 1u << ca;
      uint32_t nbm = 0;
      if (lane == ra) nbm = (bit << 1) | (bit >> 1);
      if (lane == ra - 1 || lane == ra + 1) nbm = bit;
      nbm &= g.full_l1;
      uint32_t multi_opp, alive_opp, multi_mine;
      analyze<R>(opp, mine, g.full_l1 & ~(mine | opp), lc, sc, rows5, lane, multi_opp, alive_opp, multi_mine);
      const uint32_t noair = opp & ~alive_opp;   // every opponent stone in a liberty-less group
      dead = nbm & noair;                        // ... of which only the groups touching the stone die (:169-171)
#pragma unroll 1
      for (int it = 0; it < R * R; ++it) {
        uint32_t up = __shfl_up(dead, 1), dn = __shfl_down(dead, 1);
        if (lane == 0) up = 0;
        uint32_t grown = dead | (((dead << 1) | (dead >> 1) | up | dn) & noair);
        const bool ch = grown != dead;
        dead = grown;
        if (__ballot(ch) == 0) break;
      }
      opp &= ~dead;
    }
    if (__ballot(dead != 0)) {
      black = pl ? opp : mine;
      white = pl ? mine : opp;
      WAVE_SYNC();
      row_to_plane<R>(iobuf + mi, black, N, lane);
      row_to_plane<R>(iobuf + mi + g.P, white, N, lane);
      WAVE_SYNC();
      stage_out(gs, 2 * g.P, iobuf, lane);
    }
    if (killed) {
      uint8_t *gk = killed + b * (int64_t)g.P;
      WAVE_SYNC();
      row_to_plane<R>(iobuf + ((uintptr_t)gk & 15u), dead, N, lane);
      WAVE_SYNC();
      stage_out(gk, g.P, iobuf, lane);
    }
  }
}

// k-th (0-based) valid action in ascending index order; valid = L1 rows of playable points; k == count -> pass
__device__ __forceinline__ int pick_action(uint32_t valid, uint32_t k, const Geo &g, int lane) {
  int cnt = __popc(valid);
  int incl = cnt;  // inclusive prefix over lanes 0..31 (rows live in lanes < 32)
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) {
    int t = __shfl_up(incl, off);
    if ((lane & 31) >= off) incl += t;
  }
  uint64_t hit = __ballot(lane < 32 && (uint32_t)incl > k);
  if (hit == 0) return g.P;  // pass
  int r = __ffsll((unsigned long long)hit) - 1;
  uint32_t row = __builtin_amdgcn_readlane(valid, r);
  uint32_t before = (uint32_t)__builtin_amdgcn_readlane(incl, r) - (uint32_t)__popc(row);
  uint32_t t = k - before;
  for (uint32_t i = 0; i < t; ++i) row &= row - 1;
  return r * g.N + (__ffs(row) - 1);
}

template <int R>
__global__ __launch_bounds__(kWave) void k_rollout(uint8_t *__restrict__ states, uint64_t *__restrict__ rng,
                                                   int32_t *__restrict__ last_actions, int64_t *__restrict__ steps_done,
                                                   int64_t B, int N, uint32_t inv, int plies, int auto_reset) {
  __shared__ __attribute__((aligned(16))) uint8_t iobuf[Cfg<R>::kIoBytes];
  __shared__ __attribute__((aligned(16))) uint32_t sc[kWave * Cfg<R>::kRowStride];
  __shared__ __attribute__((aligned(16))) uint32_t rows5[160];
  const int lane = threadIdx.x;
  Geo g;
  g.N = N; g.P = N * N; g.inv = inv;
  g.full_l1 = lane < N ? (1u << N) - 1u : 0u;
  const int S = 6 * g.P;
  const LaneClass lc = make_lane_class(lane);
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    uint8_t *gs = states + b * (int64_t)S;
    uint32_t flags = load_flags(gs, g.P, 0, lane);
    WAVE_SYNC();
    const uint32_t mi = stage_in(gs, 4 * g.P, iobuf, lane);
    WAVE_SYNC();
    uint32_t black = plane_to_row<R>(iobuf + mi, N, lane);
    uint32_t white = plane_to_row<R>(iobuf + mi + g.P, N, lane);
    uint32_t invalid = plane_to_row<R>(iobuf + mi + 3 * g.P, N, lane);
    int turn = flags & 1u, passed = (flags >> 2) & 1u, done = (flags >> 3) & 1u;
    uint64_t x = uniform64(rng[b]);
    int last = -1, played = 0;
#pragma unroll 1
    for (int t = 0; t < plies; ++t) {
      if (done) {
        if (!auto_reset) break;
        black = white = invalid = 0;
        turn = passed = done = 0;
      }
      uint32_t valid = g.full_l1 & ~invalid;
      int cnt = __popc(valid);
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
      cnt = __builtin_amdgcn_readfirstlane(cnt);
      uint64_t u = splitmix_next(x);
      uint32_t k = (uint32_t)(((u >> 32) * (uint64_t)(cnt + 1)) >> 32);
      int a = pick_action(valid, k, g, lane);
      uint32_t mine = turn ? white : black, opp = turn ? black : white;
      invalid = step_core<R>(mine, opp, a, g, lc, sc, rows5, lane);
      black = turn ? opp : mine;
      white = turn ? mine : opp;
      if (a == g.P) { if (passed) done = 1; passed = 1; } else passed = 0;
      turn ^= 1;
      last = a;
      ++played;
    }
    PlaneBytes pb;
    pb.turn = (uint8_t)turn; pb.passed = (uint8_t)passed; pb.done = (uint8_t)done;
    if (played) {
      emit_board<R>(iobuf + mi, black, white, invalid, pb, g, lane);
      stage_out(gs, S, iobuf, lane);
    }
    if (lane == 0) {
      rng[b] = x;
      if (last_actions) last_actions[b] = last;
      if (steps_done) steps_done[b] += played;
    }
  }
}

template <int R>
__global__ __launch_bounds__(kWave) void k_sample(const uint8_t *__restrict__ states, uint64_t *__restrict__ rng,
                                                  int32_t *__restrict__ actions, int64_t B, int N) {
  __shared__ __attribute__((aligned(16))) uint8_t iobuf[Cfg<R>::kIoBytes];
  const int lane = threadIdx.x;
  Geo g;
  g.N = N; g.P = N * N; g.inv = 0;
  g.full_l1 = lane < N ? (1u << N) - 1u : 0u;
  const int S = 6 * g.P;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    const uint8_t *gs = states + b * (int64_t)S;
    uint32_t flags = load_flags(gs, g.P, 0, lane);
    WAVE_SYNC();
    const uint32_t mi = stage_in(gs + 3 * g.P, g.P, iobuf, lane);
    WAVE_SYNC();
    uint32_t invalid = plane_to_row<R>(iobuf + mi, N, lane);
    if (flags & 8u) invalid = 0;  // gogame.invalid_moves: zeros once the game ended (gogame.py:155-156)
    uint32_t valid = g.full_l1 & ~invalid;
    int cnt = __popc(valid);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    cnt = __builtin_amdgcn_readfirstlane(cnt);
    uint64_t x = uniform64(rng[b]);
    uint64_t u = splitmix_next(x);
    uint32_t k = (uint32_t)(((u >> 32) * (uint64_t)(cnt + 1)) >> 32);
    int a = pick_action(valid, k, g, lane);
    if (lane == 0) {
      rng[b] = x;
      actions[b] = a;
    }
  }
}

}  // namespace gg
