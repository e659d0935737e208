// gg_ws.h - policy-weighted action sampling on the device: gogame.random_weighted_action / random_action
// (gym_go/gogame.py:385-404) for every game of a batch, with the per-game generator of the rollout kernels.
//
// The reference L1-normalises the move weights in float64 and draws with NumPy's global generator.  Floating-point sums
// depend on their order, so the build defines an EXACT integer form that the device and the CPU restatement the
// tests check it against evaluate identically (spelled out in include/gymgo_amd.h, gg_batch_sample_weighted):
//   1. v[a] = the float32 weight clamped to [+0, FLT_MAX] on its bit pattern (sign bit set, -NaN included -> 0; +NaN / +inf -> FLT_MAX), and 0
//      where plane 3 marks the point invalid (":387 assumes all invalid moves have weight 0" - enforced here); the pass is
//      never masked, a finished game masks nothing (gogame.invalid_moves, gym_go/gogame.py:155-156);
//   2. E = max(biased exponent of max v, 24), S = 2^(148 - E), q[a] = trunc(v[a] * S) < 2^22 - an exact power-of-two
//      scaling: the weights as 22-bit fixed point relative to the largest one;
//   3. T = sum q; one draw u of the game's generator, k = floor((u >> 32) T / 2^32);
//   4. inverse CDF in the interleaved action order a = i + 16 j (i outer, j inner): P(a) = q[a] / T.
// T == 0 gives -1 (np.random.choice raises for an all-zero vector); the generator advances once per draw.
//
// Layout: ONE DPP ROW (16 lanes) PER BOARD, four boards per wavefront pass; lane i of a row owns the actions i + 16 j,
// so every global load of a row is one contiguous 64-byte segment, the row-wide maximum / sum are four `row_ror`
// rotations and the prefix four `row_shr` shifts: 6 VALU per weight + ~40 per board.
#pragma once
#include "gg_common.h"
#include "gg_v4.h"

namespace gg {

template <int R>
struct WsCfg {
  static constexpr int kA = R * R + 1;
  static constexpr int kNJ = (kA + 15) / 16;        // actions per lane
  static constexpr int kVW = (kA + 31) / 32;        // words of a board's valid-action bit-string
};

// sum / max over the 16 lanes of a DPP row, result in every lane (row_ror:n = 0x120 + n)
__device__ __forceinline__ uint32_t row_sum16(uint32_t x) {
  x += dpp0<0x121>(x); x += dpp0<0x122>(x); x += dpp0<0x124>(x); return x + dpp0<0x128>(x);
}
__device__ __forceinline__ uint32_t row_max16(uint32_t x) {
  uint32_t y = dpp0<0x121>(x); x = x > y ? x : y;
  y = dpp0<0x122>(x); x = x > y ? x : y;
  y = dpp0<0x124>(x); x = x > y ? x : y;
  y = dpp0<0x128>(x); return x > y ? x : y;
}
// inclusive prefix sum inside a DPP row (row_shr:n = 0x110 + n; lanes shifted in from outside the row read 0)
__device__ __forceinline__ uint32_t row_scan16(uint32_t x) {
  x += dpp0<0x111>(x); x += dpp0<0x112>(x); x += dpp0<0x114>(x); return x + dpp0<0x118>(x);
}

// One draw for the board of this lane's row.  bits[j] = weight bits of action i + 16 j (anything beyond the action
// range must come with vm[j] = 0), vm[j] = 0 / ~0: the action may be played.  Returns the action in the ONE lane of
// the row that holds it, -1 in lane 0 of a row whose weights are all zero, -2 everywhere else.
template <int NJ>
__device__ __forceinline__ int wsample_row(const uint32_t (&bits)[NJ], const uint32_t (&vm)[NJ], uint32_t uhi, int lane) {
  const int i = lane & 15;
  uint32_t v[NJ], mx = 0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    int32_t s;   // clamp to [+0, FLT_MAX] on the bit pattern: one v_med3_i32
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(s) : "v"((int32_t)bits[j]), "v"(0x7F7FFFFF));
    v[j] = (uint32_t)s & vm[j];
    mx = v[j] > mx ? v[j] : mx;
  }
  mx = row_max16(mx);
  uint32_t E = mx >> 23;
  E = E < 24u ? 24u : E;
  const float S = __uint_as_float((275u - E) << 23);
  uint32_t p[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const uint32_t q = (uint32_t)(__uint_as_float(v[j]) * S);   // exact product (power of two), truncation
    p[j] = q + (j ? p[j - 1] : 0u);
  }
  const uint32_t own = p[NJ - 1];
  const uint32_t incl = row_scan16(own), T = row_sum16(own), excl = incl - own;
  const uint32_t k = (uint32_t)(((uint64_t)uhi * (uint64_t)T) >> 32);
  const bool hit = T != 0u && k >= excl && k < incl;
  // position inside the lane's run: how many of its running sums the draw has passed.  (k - excl once instead of excl + p[j]
  // per weight; in a lane that is not `hit` the difference wraps and the count is not used)
  const uint32_t kk = k - excl;
  uint32_t cnt = 0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) cnt += (p[j] <= kk) ? 1u : 0u;
  if (hit) return i + 16 * (int)cnt;
  return (T == 0u && i == 0) ? -1 : -2;
}

// weights of the row's board -> registers as float32 bit patterns (contiguous 64- / 32-byte segments per load
// instruction and row).  wt: GG_W_F32 float32, GG_W_BF16 bfloat16, GG_W_F16 float16 - the 16-bit forms are widened
// exactly, so the draw is the one of the float32 weights with the same values at half the HBM traffic.  The loads are
// UNCONDITIONAL on a clamped index (the caller hands over a readable row also for an absent board; what lies beyond the
// action range is masked by vm[]): a load under a per-element condition is a branch whose join waits for the data, i.e.
// NJ dependent memory round trips instead of NJ loads in flight (measured: 64 -> 26 us per 65 536 boards).
// AMIN: a compile-time lower bound of A (the exact value when the board fills its row capacity): elements below it
// are addressed as base + constant (one address computation, immediate offsets), only the rest through a clamped index.
template <int NJ, int AMIN = 5>
__device__ __forceinline__ void wload_row(const void *__restrict__ wbase, int64_t board, int wt, int A, int lane, uint32_t (&bits)[NJ]) {
  const int i = lane & 15;
  if (wt == GG_W_F32) {
    const float *wrow = reinterpret_cast<const float *>(wbase) + board * (int64_t)A;
    const float *wl = wrow + i;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int a = i + 16 * j;
      bits[j] = __float_as_uint(16 * j + 15 < AMIN ? wl[16 * j] : wrow[a < A ? a : A - 1]);
    }
  } else {
    const uint16_t *wrow = reinterpret_cast<const uint16_t *>(wbase) + board * (int64_t)A;
    const uint16_t *wl = wrow + i;
    uint32_t h[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int a = i + 16 * j;
      h[j] = 16 * j + 15 < AMIN ? wl[16 * j] : wrow[a < A ? a : A - 1];
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (wt == GG_W_BF16) {
        bits[j] = h[j] << 16;
      } else {
        _Float16 x;
        const uint16_t hv = (uint16_t)h[j];
        __builtin_memcpy(&x, &hv, 2);
        bits[j] = __float_as_uint((float)x);
      }
    }
  }
}

// vm[j] from a valid-action bit-string in LDS (bit a = action a may be played; bits beyond the action range clear)
template <int NJ>
__device__ __forceinline__ void wmask_from_bits(const uint32_t *vw, int lane, uint32_t (&vm)[NJ]) {
  const int i = lane & 15;
#pragma unroll
  for (int j = 0; j < NJ; ++j) vm[j] = (uint32_t)__builtin_amdgcn_sbfe((int)vw[j >> 1], 16 * (j & 1) + i, 1);
}

// gogame.random_weighted_action for every game, byte-plane states (gg_batch_sample_weighted): four boards per wave.
// The validity of an action is read straight from plane 3 (one byte per action, the same interleaved pattern as the
// weights); states == nullptr: nothing is masked.
template <int R>
__global__ __launch_bounds__(kWave) void k_sample_weighted(const uint8_t *__restrict__ states, const void *__restrict__ weights, int wt,
                                                           uint64_t *__restrict__ rng, int32_t *__restrict__ actions,
                                                           int64_t B, int N) {
  constexpr int NJ = WsCfg<R>::kNJ;
  const int lane = threadIdx.x, row = lane >> 4, i = lane & 15;
  const int P = N * N, A = P + 1, S = 6 * P;
  for (int64_t b0 = (int64_t)blockIdx.x * 4; b0 < B; b0 += (int64_t)gridDim.x * 4) {
    const bool on = b0 + row < B;
    const int64_t b = on ? b0 + row : B - 1;
    uint32_t bits[NJ], vm[NJ];
    wload_row<NJ>(weights, b, wt, A, lane, bits);
    uint64_t x = rng[b];
    if (states) {
      const uint8_t *gs = states + b * (int64_t)S;
      const uint32_t ended = gs[5 * P];
      uint32_t inv[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {   // (unconditional loads on a clamped index, see wload_row)
        const int a = i + 16 * j;
        inv[j] = gs[3 * P + (a < P ? a : P - 1)];
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int a = i + 16 * j;
        vm[j] = (on && a < A && (a == P || ended || inv[j] == 0u)) ? ~0u : 0u;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j) vm[j] = (on && i + 16 * j < A) ? ~0u : 0u;
    }
    const uint64_t u = splitmix_next(x);
    const int a = wsample_row<NJ>(bits, vm, (uint32_t)(u >> 32), lane);
    if (on && a != -2) actions[b] = a;
    if (on && i == 0) rng[b] = x;
  }
}

// The same on TRACKED (W = 5 N + 1 words per board) or PACKED (W = 3 N + 1) boards: the invalid-move rows are words
// 2 N .. 3 N - 1 of a board, the flag word (bit 2: game over) its last one.  The rows become a valid-action bit-string in LDS.
template <int R>
__global__ __launch_bounds__(kWave) void k_sample_weighted_rows(const uint32_t *__restrict__ boards, int W,
                                                                const void *__restrict__ weights, int wt, uint64_t *__restrict__ rng,
                                                                int32_t *__restrict__ actions, int64_t B, int N) {
  constexpr int NJ = WsCfg<R>::kNJ, VW = (WsCfg<R>::kVW + 1) & ~1;
  __shared__ uint32_t vbits[4][VW + 2];
  const int lane = threadIdx.x, row = lane >> 4, i = lane & 15;
  const int P = N * N, A = P + 1;
  const uint32_t full = (1u << N) - 1u;
  for (int64_t b0 = (int64_t)blockIdx.x * 4; b0 < B; b0 += (int64_t)gridDim.x * 4) {
    const bool on = b0 + row < B;
    const int64_t b = on ? b0 + row : B - 1;
    uint32_t bits[NJ], vm[NJ];
    wload_row<NJ>(weights, b, wt, A, lane, bits);
    uint64_t x = rng[b];
    const uint32_t *gb = boards + b * (int64_t)W;
    const uint32_t ended = (gb[W - 1] >> 2) & 1u;
    const uint32_t r0v = gb[2 * N + (i < N ? i : N - 1)], r1v = gb[2 * N + (i + 16 < N ? i + 16 : N - 1)];
    WAVE_SYNC();
    for (int w = i; w < VW + 2; w += 16) vbits[row][w] = 0;
    WAVE_SYNC();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = i + 16 * h;
      if (r < N) {
        const uint32_t ok = full & ~((h ? r1v : r0v) & (ended ? 0u : ~0u));
        const uint32_t q = (uint32_t)(r * N);
        const uint64_t sh = (uint64_t)ok << (q & 31u);
        atomicOr(&vbits[row][q >> 5], (uint32_t)sh);
        if ((uint32_t)(sh >> 32)) atomicOr(&vbits[row][(q >> 5) + 1], (uint32_t)(sh >> 32));
      }
    }
    if (i == 0) atomicOr(&vbits[row][P >> 5], 1u << (P & 31));   // the pass
    WAVE_SYNC();
    wmask_from_bits<NJ>(vbits[row], lane, vm);
    const uint64_t u = splitmix_next(x);
    const int a = wsample_row<NJ>(bits, vm, (uint32_t)(u >> 32), lane);
    if (on && a != -2) actions[b] = a;
    if (on && i == 0) rng[b] = x;
  }
}

}  // namespace gg
