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

// All eight views of every game (orient == nullptr), boards of up to MAXB staged bytes: the SCATTER form.  The gather above
// reads every source byte eight times (once per view) through one-byte LDS instructions: 8 x (C reads + C writes + 1 table
// read) per point, and the kernel is bound by the LDS instruction rate, not by HBM (2.8 TB/s moved).  Here a lane reads the C
// bytes of a source point ONCE and writes them to the point's place in each of the eight views - the eight destinations are
// the combinations of (r | N-1-r, c | N-1-c) and their transposes: four products and eight adds - so the byte reads and the
// table reads are shared.  Four output boards at a time are assembled in LDS (views 0-3, then 4-7: (1 + C + 4 C) LDS
// instructions per point and half instead of 4 (1 + 2 C)) and leave as aligned 16-byte vectors.
template <int MAXB>
__global__ __launch_bounds__(kWave) void k_symmetry_all8(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int64_t B, int C, int N) {
  constexpr int VH = 4;                                   // views assembled at a time (two halves: LDS for ~12 waves per CU instead of 7)
  constexpr int NV = (MAXB + 30 + 15) / 16 / kWave + 1;   // 16-byte vectors per lane of a staged board
  static_assert(NV <= 3, "vectors per lane of a staged board");
  __shared__ __attribute__((aligned(16))) uint8_t src[MAXB + 32];
  __shared__ __attribute__((aligned(16))) uint8_t dst[VH][MAXB + 32];
  __shared__ uint16_t rc[GG_MAX_BOARD * GG_MAX_BOARD];   // (row << 8) | column of point q
  const int lane = threadIdx.x;
  const int P = N * N, S = C * P;
  for (int q = lane; q < P; q += kWave) {
    const int r = q / N;
    rc[q] = (uint16_t)((r << 8) | (q - r * N));
  }
  // the source board of the NEXT iteration is fetched (aligned superset, into registers) before this one is worked on:
  // a wave's load -> scatter -> store chain is a few microseconds long and the occupancy is low, so the load is hidden
  uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0, n2 = n0;
  uint32_t nmi = 0;
  int nnv = 0;
#define GG_SYM_FETCH(BB)                                                                  \
  do {                                                                                     \
    const uint8_t *g_ = in + (BB) * (int64_t)S;                                            \
    nmi = (uint32_t)((uintptr_t)g_ & 15u);                                                 \
    nnv = (int)(nmi + (uint32_t)S + 15u) >> 4;                                             \
    const uint4 *gv_ = reinterpret_cast<const uint4 *>(g_ - nmi);                          \
    n0 = gv_[lane < nnv ? lane : nnv - 1];                                                 \
    if (NV > 1) n1 = gv_[lane + 64 < nnv ? lane + 64 : nnv - 1];                           \
    if (NV > 2) n2 = gv_[lane + 128 < nnv ? lane + 128 : nnv - 1];                         \
  } while (0)
  if ((int64_t)blockIdx.x < B) GG_SYM_FETCH((int64_t)blockIdx.x);
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    WAVE_SYNC();   // (the previous board's scatter has read src, its stage_out has read dst; rc is written)
    const uint32_t mi = nmi;
    {
      uint4 *sv = reinterpret_cast<uint4 *>(src);
      if (lane < nnv) sv[lane] = n0;
      if (NV > 1 && lane + 64 < nnv) sv[lane + 64] = n1;
      if (NV > 2 && lane + 128 < nnv) sv[lane + 128] = n2;
    }
    if (b + gridDim.x < B) GG_SYM_FETCH(b + gridDim.x);
    uint8_t *g0 = out + b * 8 * (int64_t)S;
#pragma unroll 1
    for (int vh = 0; vh < 8; vh += VH) {
      WAVE_SYNC();
      for (int q = lane; q < P; q += kWave) {
        const int w = rc[q];
        const int r = w >> 8, c = w & 0xFF, rr = N - 1 - r, cc = N - 1 - c;
        // where the source point (r, c) lands in view o (the inverse of sym_source): o & 4 transposes the flipped point;
        // views 0-3: (r | rr) N + (c | cc), views 4-7: (cc | c) N + (r | rr)
        const int a0 = vh ? cc * N : r * N, a1 = vh ? c * N : rr * N, e0 = vh ? r : c, e1 = vh ? rr : cc;
        const int d[VH] = {a0 + e0, vh ? a1 + e0 : a0 + e1, vh ? a0 + e1 : a1 + e0, a1 + e1};
        const uint8_t *sp = src + mi + q;
        uint8_t *dp[VH];
#pragma unroll
        for (int v = 0; v < VH; ++v) dp[v] = dst[v] + (uint32_t)((uintptr_t)(g0 + (vh + v) * (int64_t)S) & 15u) + d[v];
        for (int ch = 0; ch < C; ++ch) {
          const uint8_t x = sp[ch * P];
#pragma unroll
          for (int v = 0; v < VH; ++v) dp[v][ch * P] = x;
        }
      }
      WAVE_SYNC();
#pragma unroll 1
      for (int v = 0; v < VH; ++v) stage_out(g0 + (vh + v) * (int64_t)S, S, dst[v], lane);
    }
  }
#undef GG_SYM_FETCH
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
