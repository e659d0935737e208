// gg_aux.h - the two stand-alone single-board kernels (one wavefront per board, L1 "row per lane" only): the sampler
// of gg_batch_sample_actions and the capture resolution of gg_batch_update_pieces.  Neither is on the bench path; both
// exist because the reference exposes the operation on its own (gym_go/envs/go_env.py:78-81, gym_go/state_utils.py:159-211).
#pragma once
#include "gg_common.h"

namespace gg {

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

// k-th (0-based) valid action in ascending index order; valid = L1 rows of playable points; k >= count -> pass (P)
__device__ __forceinline__ int pick_action(uint32_t valid, uint32_t k, int N, int P, int lane) {
  int incl = __popc(valid);  // inclusive prefix over lanes 0..31 (rows live in lanes < 32)
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) {
    int t = __shfl_up(incl, off);
    if ((lane & 31) >= off) incl += t;
  }
  uint64_t hit = __ballot(lane < 32 && (uint32_t)incl > k);
  if (hit == 0) return P;
  int r = __ffsll((unsigned long long)hit) - 1;
  uint32_t row = __builtin_amdgcn_readlane(valid, r);
  uint32_t before = (uint32_t)__builtin_amdgcn_readlane(incl, r) - (uint32_t)__popc(row);
  uint32_t t = k - before;
  for (uint32_t i = 0; i < t; ++i) row &= row - 1;
  return r * N + (__ffs(row) - 1);
}

// GoEnv.uniform_random_action (gym_go/envs/go_env.py:78-81) for every game: one draw of the per-game generator, the
// k-th valid action of plane 3 (every action once the game has ended: gogame.invalid_moves, gym_go/gogame.py:155-156).
template <int R>
__global__ __launch_bounds__(kWave) void k_sample(const uint8_t *__restrict__ states, uint64_t *__restrict__ rng,
                                                  int32_t *__restrict__ actions, int64_t B, int N) {
  __shared__ __attribute__((aligned(16))) uint8_t iobuf[Cfg<R>::kIoBytes];
  const int lane = threadIdx.x;
  const int P = N * N, S = 6 * P;
  const uint32_t full_l1 = lane < N ? (1u << N) - 1u : 0u;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    const uint8_t *gs = states + b * (int64_t)S;
    uint32_t flags = load_flags(gs, P, 0, lane);
    WAVE_SYNC();
    const uint32_t mi = stage_in(gs + 3 * P, P, iobuf, lane);
    WAVE_SYNC();
    uint32_t invalid = plane_to_row<R>(iobuf + mi, N, lane);
    if (flags & 8u) invalid = 0;
    uint32_t valid = full_l1 & ~invalid;
    int cnt = __popc(valid);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    cnt = __builtin_amdgcn_readfirstlane(cnt);
    uint64_t x = uniform64(rng[b]);
    uint64_t u = splitmix_next(x);
    uint32_t k = (uint32_t)(((u >> 32) * (uint64_t)(cnt + 1)) >> 32);
    int a = pick_action(valid, k, N, P, lane);
    if (lane == 0) {
      rng[b] = x;
      actions[b] = a;
    }
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

}  // namespace gg
