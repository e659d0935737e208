// gg_sym.h - batched symmetry augmentation on the device: gogame.all_symmetries / random_symmetry
// (gym_go/gogame.py:340-382) for whole batches, one orientation per game or all eight.
//
// Orientation o in 0..7 exactly as the reference composes it (:349-357, :373-381): bit 0 = flip the columns (np.flip axis
// 2), THEN bit 1 = flip the rows (axis 1), THEN bit 2 = rotate by 90 degrees (np.rot90 over axes (1, 2):
// out[r][c] = x[c][N-1-r]).  Composed: out[r][c] = in[sr][sc] with (r', c') = bit 2 ? (c, N-1-r) : (r, c),
// sr = bit 1 ? N-1-r' : r', sc = bit 0 ? N-1-c' : c'.
//
// Two forms: byte planes uint8 [B][C][N][N] (any C: states, observations, per-point targets), staged through LDS with
// aligned vector accesses only; and row-mask boards (packed 3 N + 1 / tracked 5 N + 1 words, gg_batch_pack_states /
// gg_batch_track_states), where a column flip is a bit reversal of the row masks, a row flip a lane permutation and
// the rotation a 32 x 32 bit-matrix transpose across the lanes of a half-wave - liberty classes and the invalid-move rows (ko point included) are
// geometric, so a transformed tracked board is a valid tracked board.
#pragma once
#include "gg_common.h"
#include "gymgo_amd.h"

namespace gg {

__device__ __forceinline__ void sym_source(int o, int N, int r, int c, int &sr, int &sc) {
  int r1 = r, c1 = c;
  if (o & 4) { r1 = c; c1 = N - 1 - r; }
  sr = (o & 2) ? N - 1 - r1 : r1;
  sc = (o & 1) ? N - 1 - c1 : c1;
}

// byte planes: one wave per game.  orient == nullptr: all eight views, out is [B][8][C][N][N].
// The (row, column) of every point is tabulated once per workgroup (the only divisions); per view a lane then turns its
// points into source offsets with a handful of integer ops and moves one byte per plane: ~6 instructions per byte moved
// instead of ~30 with the index arithmetic per byte (65 536 x [6,19,19]: 170 -> see DESIGN us).
template <int MAXB>   // staged bytes per board (C * N * N <= MAXB)
__global__ __launch_bounds__(kWave) void k_symmetry_bytes(const uint8_t *__restrict__ in, const int32_t *__restrict__ orient,
                                                          uint8_t *__restrict__ out, int64_t B, int C, int N) {
  __shared__ __attribute__((aligned(16))) uint8_t src[MAXB + 32];
  __shared__ __attribute__((aligned(16))) uint8_t dst[MAXB + 32];
  __shared__ uint16_t rc[GG_MAX_BOARD * GG_MAX_BOARD];   // (row << 8) | column of point q
  const int lane = threadIdx.x;
  const int P = N * N, S = C * P;
  const int views = orient ? 1 : 8;
  for (int q = lane; q < P; q += kWave) {
    const int r = q / N;
    rc[q] = (uint16_t)((r << 8) | (q - r * N));
  }
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    WAVE_SYNC();
    const uint32_t mi = stage_in(in + b * (int64_t)S, S, src, lane);
    WAVE_SYNC();
    for (int v = 0; v < views; ++v) {
      const int o = orient ? (orient[b] & 7) : v;
      uint8_t *g = out + (b * views + v) * (int64_t)S;
      const uint32_t mo = (uint32_t)((uintptr_t)g & 15u);
      WAVE_SYNC();
      for (int q = lane; q < P; q += kWave) {
        const int w = rc[q];
        int sr, sc;
        sym_source(o, N, w >> 8, w & 0xFF, sr, sc);
        const uint8_t *sp = src + mi + sr * N + sc;
        uint8_t *dp = dst + mo + q;
        for (int ch = 0; ch < C; ++ch) dp[ch * P] = sp[ch * P];
      }
      WAVE_SYNC();
      stage_out(g, S, dst, lane);
    }
  }
}

// row-mask boards: one board per 32-lane half (lane r of the half = row r of every plane).
// planes = 3 (packed) or 5 (tracked); the last word of a board (flags) is copied.
__global__ __launch_bounds__(kWave) void k_symmetry_rows(const uint32_t *__restrict__ in, const int32_t *__restrict__ orient,
                                                         uint32_t *__restrict__ out, int64_t B, int N, int planes) {
  const int lane = threadIdx.x, h = lane >> 5, hl = lane & 31;
  const int W = planes * N + 1;
  const int views = orient ? 1 : 8;
  const int64_t npairs = (B + 1) >> 1;
  for (int64_t p = blockIdx.x; p < npairs; p += gridDim.x) {
    const bool on = 2 * p + h < B;
    const int64_t b = on ? 2 * p + h : B - 1;
    const uint32_t *gi = in + b * (int64_t)W;
    const uint32_t fw = gi[W - 1];
    for (int pl = 0; pl < planes; ++pl) {
      const uint32_t x = hl < N ? gi[pl * N + hl] : 0u;
      // the transpose of the plane, once: bit c of row r <- bit r of row c.  Five block-swap stages over the 32 lanes of
      // the half (lane = row of a 32 x 32 bit matrix): at distance j the lanes without bit j take their partner's low
      // column blocks into their high ones, the lanes with bit j the partner's high blocks into their low ones
      // (N ballots + N selects per plane before: 48 -> 26.6 us per 65 536 tracked boards)
      uint32_t xt = x;
#define GG_TSTAGE(J, LOWM)                                                       \
      {                                                                          \
        const uint32_t y_ = (uint32_t)__shfl_xor((int)xt, J);                    \
        const uint32_t up_ = (xt & (LOWM)) | ((y_ & (LOWM)) << (J));      /* (lane & J) == 0 */ \
        const uint32_t dn_ = (xt & ~(LOWM)) | ((y_ & ~(LOWM)) >> (J));    /* (lane & J) != 0 */ \
        xt = (hl & (J)) ? dn_ : up_;                                             \
      }
      GG_TSTAGE(16, 0x0000FFFFu)
      GG_TSTAGE(8, 0x00FF00FFu)
      GG_TSTAGE(4, 0x0F0F0F0Fu)
      GG_TSTAGE(2, 0x33333333u)
      GG_TSTAGE(1, 0x55555555u)
#undef GG_TSTAGE
      for (int v = 0; v < views; ++v) {
        const int o = orient ? (orient[b] & 7) : v;
        // out[r][c] = y[sr][sc]; with the rotation, (r', c') = (c, N-1-r): out[r] bit c = X[c'' ...]: work on the
        // transposed plane xt instead (xt[r] bit c = x[c] bit r):
        //   no rotation:  out[r] bit c = x[R(r)] bit C(c)                       R, C = optional reversals
        //   rotation:     out[r] bit c = x[R(c)] bit C(N-1-r) = xt[C(N-1-r)] bit R(c)
        uint32_t y;
        if (o & 4) {
          const int srow = (o & 1) ? hl : N - 1 - hl;            // C(N-1-r): reversed again when the columns flip
          const uint32_t t = (uint32_t)__shfl((int)xt, (lane & 32) + (srow >= 0 && srow < N ? srow : 0));
          y = (o & 2) ? (__brev(t) >> (32 - N)) : t;             // R(c): the row flip becomes a reversal of the bits
        } else {
          const int srow = (o & 2) ? N - 1 - hl : hl;
          const uint32_t t = (uint32_t)__shfl((int)x, (lane & 32) + (srow >= 0 && srow < N ? srow : 0));
          y = (o & 1) ? (__brev(t) >> (32 - N)) : t;
        }
        if (on && hl < N) out[(b * views + v) * (int64_t)W + pl * N + hl] = y;
      }
    }
    if (on && hl < views) out[(b * views + hl) * (int64_t)W + W - 1] = fw;
  }
}

}  // namespace gg
