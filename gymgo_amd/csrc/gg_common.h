// gg_common.h - building blocks shared by both kernel families of the batched Go step path (gfx950 / CDNA4).
//
// Layouts: L1 "row per lane" - lane r (< N) holds row r of a plane as a 32-bit mask; L2 "flood
// per lane" - a lane holds all rows of one colour and runs its own flood fill.  This header has the pieces that do not
// depend on how many boards share a wavefront: the wave-local LDS hand-off, aligned HBM <-> LDS staging, the
// byte-plane <-> row-mask conversions, the sampler's generator and two small whole-batch kernels.
// Reference citations are path:line relative to the reference root (huangeddie/GymGo).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gymgo_amd.h"

namespace gg {

constexpr int kWave = 64;

// 64-thread workgroups: the only "other threads" are lanes of the same wave.  DS instructions of one wave
// execute in program order, so LDS hand-offs between lanes need no s_barrier and no s_waitcnt (a
// __syncthreads() would also drain vmcnt, i.e. wait for every outstanding global store) - only the compiler
// must not move LDS accesses across the hand-off point.
#define WAVE_SYNC()                        \
  do {                                     \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_wave_barrier();       \
    asm volatile("" ::: "memory");         \
  } while (0)

template <int R>
struct Cfg {
  static constexpr int kMaxP = R * R;
  static constexpr int kRowStride = (R + 3) & ~3;               // flood scratch words per lane (16-B multiple)
  static constexpr int kIoBytes = ((6 * R * R + 15 + 15) + 15) & ~15;  // staged board + both misalignments
  static constexpr int kCellsPerLane = (R * R + kWave - 1) / kWave;
  static constexpr int kRowsPerBallotMin = kWave / R;
  static constexpr int kMaxBallots = (R + kRowsPerBallotMin - 1) / kRowsPerBallotMin;
};

// ---------------------------------------------------------------- staging: HBM <-> LDS <-> bitboards
// Boards start at arbitrary byte offsets (6 N^2 is only a multiple of 2) and rows are N bytes long, but on gfx950
// unaligned 4/8/16-byte LDS accesses are ~22x slower than aligned ones and unaligned 16-byte global accesses run
// at about half rate (tools/ubench/lds_unaligned2.hip, tools/time_align.py).  Everything below therefore touches
// HBM and LDS with naturally aligned accesses only.
struct __attribute__((aligned(16))) V16a { uint32_t w[4]; };

// HBM -> LDS copy of one board slice with ALIGNED 16-byte loads only: the aligned superset of the slice is
// fetched (the <= 30 extra bytes share a 16-byte chunk, hence a mapped page, with valid bytes) and the slice
// sits at byte offset mis = g & 15 of the LDS buffer.  Returns mis.
__device__ __forceinline__ uint32_t stage_in(const uint8_t *g, int nbytes, uint8_t *lds, int lane) {
  const uint32_t mis = (uint32_t)((uintptr_t)g & 15u);
  const uint8_t *ga = g - mis;
  const int nv = (int)(mis + nbytes + 15) >> 4;
  for (int v = lane; v < nv; v += kWave)
    *reinterpret_cast<V16a *>(lds + 16 * v) = *reinterpret_cast<const V16a *>(ga + 16 * v);
  return mis;
}

// LDS -> HBM (lds[mis + j] = byte j, mis = g & 15): aligned 16-byte stores for the covered vectors, ONE
// global_store_byte instruction (lanes 0-14 head, 16-30 tail) for the ragged edges - neighbours are never touched.
__device__ __forceinline__ void stage_out(uint8_t *g, int nbytes, const uint8_t *lds, int lane) {
  const uint32_t mis = (uint32_t)((uintptr_t)g & 15u);
  uint8_t *ga = g - mis;
  const int end = (int)mis + nbytes;
  const int v0 = mis ? 1 : 0, v1 = end >> 4;
  for (int v = v0 + lane; v < v1; v += kWave)
    *reinterpret_cast<V16a *>(ga + 16 * v) = *reinterpret_cast<const V16a *>(lds + 16 * v);
  if (v1 >= v0) {
    const int head = mis ? 16 - (int)mis : 0, tail = end & 15;
    int j = -1;
    if (lane < 16) { if (lane < head) j = lane; }
    else if (lane < 32 && lane - 16 < tail) j = nbytes - tail + (lane - 16);
    if (j >= 0) g[j] = lds[mis + j];
  } else {
    for (int i = lane; i < nbytes; i += kWave) g[i] = lds[mis + i];
  }
}

// One byte plane (P bytes of 0/1 in LDS, any byte alignment) -> L1 row mask.
// UNALIGNED 4-byte LDS accesses are ~22x slower than aligned ones on gfx950 (tools/ubench/lds_unaligned2.hip:
// 26.9 ns vs 1.24 ns per wave instruction), so lane r reads the ALIGNED dwords that cover its row, packs
// 4 cells per v_dot4_u32_u8 (weights 1,2,4,8 / 16,32,64,128) and shifts the sub-dword offset out at the end.
template <int R>
__device__ __forceinline__ uint32_t plane_to_row(const uint8_t *plane, int N, int lane) {
  constexpr int ND = ((R + 3 + 3) / 4 + 1) & ~1;  // aligned dwords covering 3 + R bytes, even count
  uint32_t row = 0;
  if (lane < N) {
    const uint8_t *p = plane + lane * N;
    const uint32_t s = (uint32_t)((uintptr_t)p & 3u);
    const uint32_t *d = reinterpret_cast<const uint32_t *>(p - s);
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < ND; k += 2) {
      uint32_t g = __builtin_amdgcn_udot4(d[k] & 0x01010101u, 0x08040201u, 0u, false);
      g = __builtin_amdgcn_udot4(d[k + 1] & 0x01010101u, 0x80402010u, g, false);
      acc |= g << (4 * k);
    }
    row = (acc >> s) & ((1u << N) - 1u);
  }
  return row;
}

// L1 row mask -> one byte plane in LDS (v1 kernels and the mask output): plain byte stores, always aligned.
template <int R>
__device__ __forceinline__ void row_to_plane(uint8_t *plane, uint32_t row, int N, int lane) {
  if (lane < N) {
    uint8_t *p = plane + lane * N;
#pragma unroll
    for (int c = 0; c < R; ++c)
      if (c < N) p[c] = (uint8_t)((row >> c) & 1u);
  }
}

// uniform plane (turn / passed / done): every byte = val
__device__ __forceinline__ void splat_plane(uint8_t *plane, uint32_t val, int P, int lane) {
  for (int i = lane; i < P; i += kWave) plane[i] = (uint8_t)val;
}

// row / column of a flat action with the host-supplied reciprocal: inv = ceil(2^16 / N), exact for a <= N*N
__device__ __forceinline__ void split_action(int a, int N, uint32_t inv, int &r, int &c) {
  r = (int)(((uint32_t)a * inv) >> 16);
  c = a - r * N;
}

// wave-uniform copy of a 64-bit value (readfirstlane returns a SIGNED int: cast before widening)
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | (uint64_t)lo;
}

// ---- sampler shared by the rollout kernels (mirrors oracle/gg_oracle.c splitmix_next / rollout_ply)
__device__ __forceinline__ uint64_t splitmix_next(uint64_t &x) {
  uint64_t z = (x += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// GoVecEnv auto-reset: games whose game-over plane is set are zeroed IN PLACE (build-side policy);
// one wave per finished board does the stores, everyone else only reads one byte.
__global__ void k_reset_finished(uint8_t *__restrict__ states, int64_t B, int N) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / kWave;
  const int64_t nwaves = (gridDim.x * (int64_t)blockDim.x) / kWave;
  const int P = N * N, S = 6 * P;
  for (int64_t b0 = wave * kWave; b0 < B; b0 += nwaves * kWave) {
    const int64_t b = b0 + lane;
    const bool done = b < B && states[b * (int64_t)S + 5 * P] != 0;
    uint64_t m = __ballot(done);
    while (m) {
      const int l = __ffsll((unsigned long long)m) - 1;
      m &= m - 1;
      uint8_t *g = states + (b0 + l) * (int64_t)S;
      const uint32_t mis = (uint32_t)((uintptr_t)g & 15u);
      uint8_t *ga = g - mis;
      const int end = (int)mis + S, v0 = mis ? 1 : 0, v1 = end >> 4;
      const V16a z = {{0u, 0u, 0u, 0u}};
      for (int v = v0 + lane; v < v1; v += kWave) *reinterpret_cast<V16a *>(ga + 16 * v) = z;
      const int head = mis ? 16 - (int)mis : 0, tail = end & 15;
      int j = -1;
      if (lane < 16) { if (lane < head) j = lane; }
      else if (lane < 32 && lane - 16 < tail) j = S - tail + (lane - 16);
      if (j >= 0) g[j] = 0;
    }
  }
}

__global__ void k_rng_seed(uint64_t *rng, uint64_t base_seed, int64_t first_game, int64_t B) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= B) return;
  uint64_t x = base_seed ^ ((uint64_t)(first_game + i) * 0xD1342543DE82EF95ull);
  splitmix_next(x);
  rng[i] = x;
}

}  // namespace gg
