// gg_aux.h - stand-alone kernels off the bench path, which exist because the reference exposes the operation on its own:
// the sampler of gg_batch_sample_actions (gym_go/envs/go_env.py:78-81) and the area scoring of gg_batch_areas
// (gym_go/gogame.py:275-300) - sixteen boards per wavefront, every load of the group issued up front - and the capture
// resolution of gg_batch_update_pieces (gym_go/state_utils.py:159-211; one wavefront per board, L1 "row per lane").
#pragma once
#include "gg_common.h"
#include "gg_v4.h"   // the quad DPP helpers

namespace gg {

// GoEnv.uniform_random_action (gym_go/envs/go_env.py:78-81) for every game: one draw of the per-game generator, the
// k-th valid action of plane 3 in ascending order, k == count: the pass (every point once the game has ended:
// gogame.invalid_moves, gym_go/gogame.py:155-156).  Sixteen boards per wavefront: all loads of the group up front (plane 3
// as aligned 16-byte vectors, one game-over byte and one generator word per board), one row of one board packed per lane,
// then the sampler of the multi-ply kernel - four lanes per board, RPL rows each, a quad scan over the row counts.
template <int R>
struct LdsSample {
  static constexpr int kRS = Cfg<R>::kRowStride, kBoards = 16;
  static constexpr int kVec = (R * R + 15 + 15) / 16 + 1;        // 16-byte vectors covering one plane at any alignment
  static constexpr int kStageBoard = kVec * 16;
  static constexpr int kPerLane = (kBoards * kVec + kWave - 1) / kWave;
  static constexpr int kRows = (kBoards * kStageBoard + 3) / 4;  // words: [16][kRS] rows of plane 3
  static constexpr int kMeta = kRows + kBoards * kRS;            // words: done[16], rng[32]
  static constexpr int kTotal = kMeta + 3 * kBoards;
};

template <int R>
__global__ __launch_bounds__(kWave) void k_sample16(const uint8_t *__restrict__ states, uint64_t *__restrict__ rng,
                                                    int32_t *__restrict__ actions, int64_t B, int N) {
  using L = LdsSample<R>;
  constexpr int RS = L::kRS, RPL = (R + 3) / 4;
  static_assert(4 * RPL <= RS, "four lanes cover the rows of a board");
  __shared__ __attribute__((aligned(16))) uint32_t lds[L::kTotal];
  uint8_t *stage = reinterpret_cast<uint8_t *>(lds);
  uint32_t *rows = lds + L::kRows, *donev = lds + L::kMeta, *rngv = lds + L::kMeta + L::kBoards;
  const int lane = threadIdx.x;
  const int P = N * N, S = 6 * P;
  const int64_t b_first = (int64_t)blockIdx.x * L::kBoards;
  uint4 v[L::kPerLane];
#pragma unroll
  for (int k = 0; k < L::kPerLane; ++k) {
    const int j = lane + kWave * k, i = j / L::kVec, w = j - i * L::kVec;
    int64_t b = b_first + i;
    if (b >= B) b = B - 1;
    const uint8_t *g = states + b * (int64_t)S + 3 * P;
    const uint32_t mis = (uint32_t)((uintptr_t)g & 15u);
    const bool need = i < L::kBoards && 16 * w < (int)mis + P;
    v[k] = need ? reinterpret_cast<const uint4 *>(g - mis)[w] : make_uint4(0u, 0u, 0u, 0u);
  }
  uint32_t done_b = 0;
  uint64_t x0 = 0;
  if (lane < L::kBoards) {
    const int64_t b = b_first + lane < B ? b_first + lane : B - 1;
    done_b = states[b * (int64_t)S + 5 * P];
    x0 = rng[b];
  }
#pragma unroll
  for (int k = 0; k < L::kPerLane; ++k) {
    const int j = lane + kWave * k, i = j / L::kVec, w = j - i * L::kVec;
    if (i < L::kBoards) *reinterpret_cast<uint4 *>(stage + i * L::kStageBoard + 16 * w) = v[k];
  }
  for (int u = lane; u < L::kBoards * RS; u += kWave) rows[u] = 0;   // rows N .. RS-1: nothing to play there
  if (lane < L::kBoards) {
    donev[lane] = done_b;
    rngv[2 * lane] = (uint32_t)x0;
    rngv[2 * lane + 1] = (uint32_t)(x0 >> 32);
  }
  WAVE_SYNC();
  for (int u = lane; u < L::kBoards * N; u += kWave) {
    const int i = u / N, r = u - i * N;
    const int64_t b = b_first + i < B ? b_first + i : B - 1;
    const uint32_t mis = (uint32_t)((uintptr_t)(states + b * (int64_t)S + 3 * P) & 15u);
    rows[i * RS + r] = plane_to_row<R>(stage + i * L::kStageBoard + mis, N, r);
  }
  WAVE_SYNC();
  // ---- four lanes per board
  const int q = lane >> 2, t = lane & 3, r0 = RPL * t;
  const uint32_t keep = donev[q] ? 0u : ~0u;
  uint32_t vr_[RPL], p[RPL];
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    const uint32_t full = (r0 + r < N) ? (1u << N) - 1u : 0u;
    vr_[r] = full & ~(rows[q * RS + r0 + r] & keep);
    p[r] = (uint32_t)__popc(vr_[r]) + (r ? p[r - 1] : 0u);
  }
  const uint32_t T = p[RPL - 1];
  // (the DPP moves are evaluated by every lane, THEN masked: inside a conditional the source lanes would be off)
  const uint32_t sh1 = dpp0<QP_SHR1>(T);
  const uint32_t x1 = T + (t >= 1 ? sh1 : 0u);
  const uint32_t sh2 = dpp0<QP_SHR2>(x1);
  const uint32_t Sx = x1 + (t >= 2 ? sh2 : 0u);
  const uint32_t n = dpp0<QP_B3>(Sx), before = Sx - T;
  uint64_t x = ((uint64_t)rngv[2 * q + 1] << 32) | rngv[2 * q];
  const uint64_t u64 = splitmix_next(x);
  const uint32_t k = (uint32_t)(((u64 >> 32) * (uint64_t)(n + 1)) >> 32);   // k == n: the pass
  const bool hit = k >= before && k < before + T;
  int rr;
  uint32_t pos;
  kth_set_bit<RPL>(vr_, p, (k - before) & 0x3FFu, rr, pos);   // (only the hit lane's result is used)
  const int64_t b = b_first + q;
  if (b < B) {
    if (k < n ? hit : t == 0) actions[b] = k < n ? (r0 + rr) * N + (int)pos : P;
    if (t == 0) rng[b] = x;
  }
}

// 4-neighbourhood dilation in L1 (lane r = row r; rows of lanes >= N are zero, callers mask the columns)
__device__ __forceinline__ uint32_t dilate_l1(uint32_t x, int lane) {
  uint32_t up = __shfl_up(x, 1), dn = __shfl_down(x, 1);
  if (lane == 0) up = 0;
  if (lane == kWave - 1) dn = 0;
  return (x << 1) | (x >> 1) | up | dn;
}

// state_utils.update_pieces / batch_update_pieces (gym_go/state_utils.py:159-211) as a stand-alone entry, with the
// reference's own inputs: adj[b][0..K) are the locations whose opponent groups are examined (flat indices, entries
// outside [0, N*N) unused - the reference passes the on-board neighbours of the stone just placed, adj_data :214-223)
// and players[b] the side that moved.  Every opponent group holding one of these locations that has NO empty point
// next to it (liberties are taken on the position as given, before any removal, exactly like `empties` at :164) is
// removed IN PLACE from planes 0/1 and reported in `killed` (0/1 per point, nullable).  Nothing else is assumed about
// the position: it need not be reachable by legal play.
template <int R>
__global__ __launch_bounds__(kWave) void k_update_pieces(uint8_t *__restrict__ states, const int32_t *__restrict__ adj,
                                                         int K, const int32_t *__restrict__ players,
                                                         uint8_t *__restrict__ killed, int64_t B, int N, uint32_t inv) {
  __shared__ __attribute__((aligned(16))) uint8_t iobuf[Cfg<R>::kIoBytes];
  const int lane = threadIdx.x;
  const int P = N * N, S = 6 * P;
  const uint32_t full_l1 = lane < N ? (1u << N) - 1u : 0u;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    uint8_t *gs = states + b * (int64_t)S;
    const int pl = __builtin_amdgcn_readfirstlane(players[b]) & 1;
    WAVE_SYNC();
    const uint32_t mi = stage_in(gs, 2 * P, iobuf, lane);
    WAVE_SYNC();
    uint32_t black = plane_to_row<R>(iobuf + mi, N, lane);
    uint32_t white = plane_to_row<R>(iobuf + mi + P, N, lane);
    uint32_t opp = pl ? black : white;
    const uint32_t empties = full_l1 & ~(black | white);
    uint32_t dead = 0, seen = 0;
#pragma unroll 1
    for (int j = 0; j < K; ++j) {
      const int a = __builtin_amdgcn_readfirstlane(adj[b * (int64_t)K + j]);
      if (a < 0 || a >= P) continue;
      int ra, ca;
      split_action(a, N, inv, ra, ca);
      uint32_t g = (lane == ra) ? (opp & ~seen & (1u << ca)) : 0u;   // np.unique(labels): a group is examined once
      if (__ballot(g != 0) == 0) continue;
#pragma unroll 1
      for (int it = 0; it < R * R; ++it) {   // the group of the seed: 4-connected opponent stones (ndimage.label, :166)
        const uint32_t grown = g | (dilate_l1(g, lane) & opp);
        const bool ch = grown != g;
        g = grown;
        if (__ballot(ch) == 0) break;
      }
      seen |= g;
      const uint32_t libs = dilate_l1(g, lane) & empties;            // :171 liberties = empties * dilation(group)
      if (__ballot(libs != 0) == 0) dead |= g;                        // :172-176
    }
    opp &= ~dead;
    if (__ballot(dead != 0)) {
      if (pl) black = opp; else white = opp;
      WAVE_SYNC();
      row_to_plane<R>(iobuf + mi, black, N, lane);
      row_to_plane<R>(iobuf + mi + P, white, N, lane);
      WAVE_SYNC();
      stage_out(gs, 2 * P, iobuf, lane);
    }
    if (killed) {
      uint8_t *gk = killed + b * (int64_t)P;
      WAVE_SYNC();
      row_to_plane<R>(iobuf + ((uintptr_t)gk & 15u), dead, N, lane);
      WAVE_SYNC();
      stage_out(gk, P, iobuf, lane);
    }
  }
}

// ---------------------------------------------------------------- gogame.areas for a batch, sixteen boards per wavefront
// gym_go/gogame.py:275-300 (Tromp-Taylor: a colour owns its stones plus the empty regions that touch only that colour).
// An empty region touches a colour iff the flood of the EMPTY points seeded next to that colour's stones covers it, so
// a board needs two floods and a wave of 64 lanes serves sixteen boards with its lower half (lane 2 s + c: board s,
// colour c) - the round-1 kernel flooded 4 lanes per wave (two boards).  Only planes 0 and 1 are read (2 N^2 bytes of
// the 6 N^2): all loads of the group are issued up front (aligned 16-byte vectors, 12 per lane at 19x19), then staged
// through LDS eight boards at a time and packed into row masks by all 64 lanes (one row of one plane each).
template <int R>
struct LdsAreas {
  static constexpr int kRS = Cfg<R>::kRowStride;
  static constexpr int kBoards = 16, kHalf = 8;
  static constexpr int kVec = (2 * R * R + 15 + 15) / 16 + 1;                     // 16-byte vectors covering planes 0 / 1 at any alignment
  static constexpr int kStageBoard = kVec * 16;
  static constexpr int kPerLane = (kHalf * kVec + kWave - 1) / kWave;             // vectors per lane and half
  // The staging area is free once the boards are packed: the converged floods (32 lanes x kRS words) and the rows the idle
  // upper half of the wave writes (another 32 x kRS) take its place.  (Round 4: they had 2 560 B of their own at 19x19 -
  // 11 264 B per workgroup, 14 waves per CU, so 512 of the 4 096 workgroups of a 65 536-board call ran as a second round.)
  static constexpr int kBlocks = 2 * kBoards * kRS;                               // words of one set of 32 flood outputs
  static constexpr int kStage = 0;                                                // bytes: kHalf * kStageBoard
  static constexpr int kFront = (kHalf * kStageBoard + 3) / 4 > 2 * kBlocks ? (kHalf * kStageBoard + 3) / 4 : 2 * kBlocks;
  static constexpr int kSc = 0;                                                   // words: [32][kRS] converged floods
  static constexpr int kIdle = kBlocks;                                           // words: the idle lanes' rows
  static constexpr int kSt = (kFront + 3) & ~3;                                   // words: [2][kBoards][kRS] stone rows
  static constexpr int kTotal = kSt + kBlocks;
};

template <int R, bool FULLN>
__global__ __launch_bounds__(kWave) void k_areas4(const uint8_t *__restrict__ states, int32_t *__restrict__ black_area,
                                                  int32_t *__restrict__ white_area, int64_t B, int N) {
  using L = LdsAreas<R>;
  if (FULLN) N = R;
  constexpr int RS = L::kRS, PL = L::kBoards * RS;
  __shared__ __attribute__((aligned(16))) uint32_t lds[L::kTotal];
  uint8_t *stage = reinterpret_cast<uint8_t *>(lds);
  uint32_t *st = lds + L::kSt, *sc = lds + L::kSc;
  const int lane = threadIdx.x;
  const int P = N * N, S = 6 * P;
  const int64_t b_first = (int64_t)blockIdx.x * L::kBoards;
  const int nv = (15 + 2 * P + 15) / 16 + 1;        // vectors fetched per board (an upper bound of the covering set)
  // ---- every load of the group, before anything waits for one
  uint4 v[2][L::kPerLane];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int k = 0; k < L::kPerLane; ++k) {
      const int j = lane + kWave * k, i = j / L::kVec, w = j - i * L::kVec;
      int64_t b = b_first + L::kHalf * h + i;
      if (b >= B) b = B - 1;
      const uint8_t *g = states + b * (int64_t)S;
      const uint32_t mis = (uint32_t)((uintptr_t)g & 15u);
      // (the covering vectors of a board: mis + 2 P bytes; a vector past them would lie past the board's planes 2..5 never)
      const bool need = i < L::kHalf && w < nv && 16 * w < (int)mis + 2 * P;
      v[h][k] = need ? reinterpret_cast<const uint4 *>(g - mis)[w] : make_uint4(0u, 0u, 0u, 0u);
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < L::kPerLane; ++k) {
      const int j = lane + kWave * k, i = j / L::kVec, w = j - i * L::kVec;
      if (i < L::kHalf) *reinterpret_cast<uint4 *>(stage + i * L::kStageBoard + 16 * w) = v[h][k];
    }
    WAVE_SYNC();
    // one row of one plane per lane and round: unit u = (board i, plane p, row r)
    for (int u = lane; u < L::kHalf * 2 * N; u += kWave) {
      const int i = u / (2 * N), rem = u - i * 2 * N, p = rem >= N ? 1 : 0, r = rem - p * N;
      int64_t b = b_first + L::kHalf * h + i;
      if (b >= B) b = B - 1;
      const uint32_t mis = (uint32_t)((uintptr_t)(states + b * (int64_t)S) & 15u);
      st[p * PL + (L::kHalf * h + i) * RS + r] = plane_to_row<R>(stage + i * L::kStageBoard + mis + p * P, N, r);
    }
  }
  if (N < RS) {   // rows N .. RS-1 of every board: no stones
    for (int u = lane; u < 2 * L::kBoards * (RS - N); u += kWave) {
      const int q = u / (RS - N), r = N + (u - q * (RS - N));
      st[q * RS + r] = 0;
    }
  }
  WAVE_SYNC();
  // ---- lane 2 s + c floods the empty points of board s from the neighbours of colour c (the idle upper half's rows go
  // to the staging area, which is free by now)
  const uint32_t cnt = areas16<R, FULLN>(st, sc, lds + L::kIdle, N, lane);
  const int s = lane >> 1, c = lane & 1;
  if (lane < 2 * L::kBoards && b_first + s < B) (c ? white_area : black_area)[b_first + s] = (int32_t)cnt;
}


}  // namespace gg
