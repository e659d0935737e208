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
#include "gg_v2.h"
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

// The fast form for what the entry point is used for most - states and observations, i.e. planes of 0 / 1 bytes: one
// wave per game, the board goes through the BIT domain.  The staged bytes are packed into row masks (v_dot4, four cells
// per instruction), every plane is transposed once (the five-stage 32 x 32 bit transpose of k_symmetry_rows), a view
// is then a row selection + optional bit reversal per row - the geometry of k_symmetry_rows - ORed into a bit-string
// of the OUTPUT range (all the views of a game are one contiguous byte range: bit i = byte i), and the bit-string leaves
// through the 8 bits -> 8 bytes table as aligned 16-byte vectors, like every board emitter of this library.  ~0.1 LDS
// instructions per output byte instead of the ~1.2 of a per-byte gather or scatter (which bound those at 2.8 - 3.9 TB/s).
// A game whose bytes are not all 0 / 1 (per-point targets ...) is detected while it is staged and takes the per-byte
// gather of k_symmetry_bytes inside this kernel: same results, round-3 speed.  The next game's bytes are fetched
// (aligned superset, registers) before the current one is worked on.  C * N <= 128 rows, C * N * N <= MAXB.
template <int MAXB>
__global__ __launch_bounds__(kWave) void k_symmetry_bits(const uint8_t *__restrict__ in, const int32_t *__restrict__ orient,
                                                         uint8_t *__restrict__ out, int64_t B, int C, int N) {
  constexpr int NV = (MAXB + 30 + 15) / 16 / kWave + 1;   // 16-byte vectors per lane of a staged board
  static_assert(NV <= 3, "vectors per lane of a staged board");
  constexpr int BSW = ((15 + 8 * MAXB + 31) / 32 + 4) & ~3;   // words of the output bit-string (eight views)
  constexpr int DSW = (MAXB + 32 + 3) / 4;                    // words of the fallback's byte buffer (aliases the bit-string)
  __shared__ __attribute__((aligned(16))) uint8_t src[MAXB + 32];
  __shared__ __attribute__((aligned(16))) uint32_t buf[BSW > DSW ? BSW : DSW];
  __shared__ uint32_t rowsX[128], rowsT[128];
  __shared__ uint2 lut[256];
  __shared__ uint16_t rc[GG_MAX_BOARD * GG_MAX_BOARD];   // fallback: (row << 8) | column of point q
  const int lane = threadIdx.x, hl = lane & 31, hh = lane >> 5;
  const int P = N * N, S = C * P, NR = C * N;
  const int views = orient ? 1 : 8;
  load_spread_lut(lut, lane);
  for (int q = lane; q < P; q += kWave) {
    const int r = q / N;
    rc[q] = (uint16_t)((r << 8) | (q - r * N));
  }
  // this lane's rows lane and lane + 64 of the C * N rows of a board: (plane, row in the plane)
  const int pA = lane / N, rA = lane - pA * N, pB = (lane + 64) / N, rB = lane + 64 - pB * N;
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
    WAVE_SYNC();   // (the previous game has read src and buf)
    const uint32_t mi = nmi;
    const int nv = nnv;
    {
      uint4 *sv = reinterpret_cast<uint4 *>(src);
      if (lane < nv) sv[lane] = n0;
      if (NV > 1 && lane + 64 < nv) sv[lane + 64] = n1;
      if (NV > 2 && lane + 128 < nv) sv[lane + 128] = n2;
    }
    if (b + gridDim.x < B) GG_SYM_FETCH(b + gridDim.x);
    WAVE_SYNC();
    // the <= 15 + 15 bytes of the aligned superset that are not the game's are cleared, so that whole vectors can be tested
    if (lane < 16) { if (lane < (int)mi) src[lane] = 0; }
    else if (lane < 32) { const int e = (int)mi + S + (lane - 16); if (e < 16 * nv) src[e] = 0; }
    WAVE_SYNC();
    uint32_t big = 0;
    {
      const uint4 *sv = reinterpret_cast<const uint4 *>(src);
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        if (lane + 64 * k < nv) { const uint4 x = sv[lane + 64 * k]; big |= (x.x | x.y | x.z | x.w) & 0xFEFEFEFEu; }
      }
    }
    const int o1 = orient ? (orient[b] & 7) : 0;
    uint8_t *g = out + b * views * (int64_t)S;
    if (__ballot(big != 0u) == 0) {
      // ---- 0 / 1 planes: rows -> masks
      WAVE_SYNC();
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int rho = lane + 64 * k;
        if (rho < NR) {
          const uint8_t *p = src + mi + rho * N;
          const uint32_t sft = (uint32_t)((uintptr_t)p & 3u);
          const uint32_t *d = reinterpret_cast<const uint32_t *>(p - sft);
          uint32_t acc = 0;
#pragma unroll
          for (int j = 0; j < 6; j += 2) {   // six aligned dwords cover 3 + 19 bytes
            uint32_t t = __builtin_amdgcn_udot4(d[j] & 0x01010101u, 0x08040201u, 0u, false);
            t = __builtin_amdgcn_udot4(d[j + 1] & 0x01010101u, 0x80402010u, t, false);
            acc |= t << (4 * j);
          }
          rowsX[rho] = (acc >> sft) & ((1u << N) - 1u);
        }
      }
      WAVE_SYNC();
      // ---- every plane transposed once (two planes per pass, one per 32-lane half)
      for (int p0 = 0; p0 < C; p0 += 2) {
        const int pl = p0 + hh;
        uint32_t xt = (pl < C && hl < N) ? rowsX[pl * N + hl] : 0u;
#define GG_TSTAGE(J, LOWM)                                                       \
        {                                                                        \
          const uint32_t y_ = (uint32_t)__shfl_xor((int)xt, J);                  \
          const uint32_t up_ = (xt & (LOWM)) | ((y_ & (LOWM)) << (J));           \
          const uint32_t dn_ = (xt & ~(LOWM)) | ((y_ & ~(LOWM)) >> (J));         \
          xt = (hl & (J)) ? dn_ : up_;                                           \
        }
        GG_TSTAGE(16, 0x0000FFFFu)
        GG_TSTAGE(8, 0x00FF00FFu)
        GG_TSTAGE(4, 0x0F0F0F0Fu)
        GG_TSTAGE(2, 0x33333333u)
        GG_TSTAGE(1, 0x55555555u)
#undef GG_TSTAGE
        if (pl < C && hl < N) rowsT[pl * N + hl] = xt;
      }
      // ---- the bit-string of the output range
      const uint32_t mo = (uint32_t)((uintptr_t)g & 15u);
      const int nbits = (int)mo + views * S;
      for (int i = lane; i < (nbits + 31) / 32 + 1; i += kWave) buf[i] = 0;
      WAVE_SYNC();
#pragma unroll 1
      for (int v = 0; v < views; ++v) {
        const int o = orient ? o1 : v;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int rho = lane + 64 * k, pl = k ? pB : pA, r = k ? rB : rA;
          if (rho < NR) {
            // (k_symmetry_rows' geometry: without the rotation out[r] bit c = x[R(r)] bit C(c); with it, on the transposed
            // plane, out[r] bit c = xt[C(N-1-r)] bit R(c))
            uint32_t y;
            if (o & 4) {
              const uint32_t t = rowsT[pl * N + ((o & 1) ? r : N - 1 - r)];
              y = (o & 2) ? (__brev(t) >> (32 - N)) : t;
            } else {
              const uint32_t t = rowsX[pl * N + ((o & 2) ? N - 1 - r : r)];
              y = (o & 1) ? (__brev(t) >> (32 - N)) : t;
            }
            if (y) {
              const uint32_t q = mo + (uint32_t)(v * S + rho * N);
              const uint64_t x = (uint64_t)y << (q & 31u);
              atomicOr(buf + (q >> 5), (uint32_t)x);
              if ((uint32_t)(x >> 32)) atomicOr(buf + (q >> 5) + 1, (uint32_t)(x >> 32));
            }
          }
        }
      }
      WAVE_SYNC();
      // ---- bits -> bytes: aligned 16-byte vectors through the table, the ragged ends as single bytes
      uint8_t *ga = g - mo;
      const int v0 = mo ? 1 : 0, v1 = nbits >> 4;
      const uint8_t *bb = reinterpret_cast<const uint8_t *>(buf);
      typedef uint32_t u4v __attribute__((ext_vector_type(4)));
      u4v *gv = reinterpret_cast<u4v *>(__builtin_assume_aligned(ga, 16));
      for (int v = v0 + lane; v < v1; v += kWave) {
        const uint2 lo = lut[bb[2 * v]], hi = lut[bb[2 * v + 1]];
        u4v ov;
        ov.x = lo.x; ov.y = lo.y; ov.z = hi.x; ov.w = hi.y;
        gv[v] = ov;
      }
      const int head = mo ? 16 - (int)mo : 0, tail = nbits & 15, total = views * S;
      int j = -1;
      if (lane < 16) { if (lane < head) j = lane; }
      else if (lane < 32 && lane - 16 < tail) j = total - tail + (lane - 16);
      if (j >= 0 && j < total) {
        const uint32_t q = mo + (uint32_t)j;
        g[j] = (uint8_t)((buf[q >> 5] >> (q & 31u)) & 1u);
      }
    } else {
      // ---- arbitrary bytes: the per-byte gather of k_symmetry_bytes, one view at a time through `buf`
      uint8_t *dst = reinterpret_cast<uint8_t *>(buf);
      for (int v = 0; v < views; ++v) {
        const int o = orient ? o1 : v;
        uint8_t *gv = g + v * (int64_t)S;
        const uint32_t mo = (uint32_t)((uintptr_t)gv & 15u);
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
        stage_out(gv, S, dst, lane);
      }
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
    // every row of the board is requested before the first one is used (planes <= 5): one memory round trip per board
    // instead of one per plane
    uint32_t xs[5];
#pragma unroll
    for (int pl = 0; pl < 5; ++pl) xs[pl] = (pl < planes && hl < N) ? gi[pl * N + hl] : 0u;
    // a view without the rotation (bit 2 of the orientation) does not need the transposed plane: a wave whose boards all
    // stay unrotated skips the five shuffle stages (65 536 tracked boards with orientations 0 .. 3: 27.0 -> 17.4 us; all
    // eight views of 8 192: 17.3 -> 15.3; random orientations 28.1 -> 27.5 - some board of nearly every wave rotates)
    const int o1 = orient ? (orient[b] & 7) : 0;
    const bool rot_any = views == 8 || __ballot(on && (o1 & 4)) != 0;
#pragma unroll
    for (int pl = 0; pl < 5; ++pl) {
      if (pl >= planes) break;
      const uint32_t x = xs[pl];
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
      if (rot_any) {
        GG_TSTAGE(16, 0x0000FFFFu)
        GG_TSTAGE(8, 0x00FF00FFu)
        GG_TSTAGE(4, 0x0F0F0F0Fu)
        GG_TSTAGE(2, 0x33333333u)
        GG_TSTAGE(1, 0x55555555u)
      }
#undef GG_TSTAGE
      for (int v = 0; v < views; ++v) {
        const int o = orient ? o1 : v;
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
