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

// ---------------------------------------------------------------- 2-cycle op spelling + the flood fill
// Instruction selection (tools/ubench/valu_rate2.hip, measured on MI355X): v_and/or/xor/add/sub/lshrrev/bitop3/mov
// issue in 2 cycles per wave64; v_bfrev, v_and_or, v_or3, v_lshl_or, v_lshlrev, v_bfi, v_bcnt, v_bfe, v_mul_u32_u24,
// v_dot4, v_readlane cost 4.  The hot loops therefore spell every 3-input boolean as v_bitop3_b32 and "<< 1" as an add.
// v_bitop3_b32 truth tables: result bit = table[(a << 2) | (b << 1) | c] with a = 0xF0, b = 0xCC, c = 0xAA
constexpr uint32_t TA = 0xF0, TB = 0xCC, TC = 0xAA;
constexpr uint32_t T_ANDOR = (TA & TB) | TC;                    // (a & b) | c
constexpr uint32_t T_SEL = (TA & TB) | (~TA & TC & 0xFF);       // a ? b : c
constexpr uint32_t T_AND_ANDN = TA & TB & (~TC & 0xFF);         // a & b & ~c
constexpr uint32_t T_OR3 = TA | TB | TC;
constexpr uint32_t T_XOR3 = TA ^ TB ^ TC;
constexpr uint32_t T_MAJ = (TA & TB) | (TC & (TA | TB));
constexpr uint32_t T_AND_OR2 = TA & (TB | TC);                  // a & (b | c)
constexpr uint32_t T_OR_AND = TA | (TB & TC);                   // a | (b & c)
#define B3(a, b, c, t) __builtin_amdgcn_bitop3_b32((a), (b), (c), (t))

__device__ __forceinline__ uint32_t shl1(uint32_t x) {  // x << 1 as a 2-cycle add (v_lshlrev_b32 costs 4)
  uint32_t r;
  asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(x));
  return r;
}

// DPP moves (GFX9 encodings): row_shr:n = 0x110 + n, row_bcast:15 = 0x142, wave_shl:1 = 0x130, wave_shr:1 = 0x138
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
}

// OR-accumulation of a compile-time-indexed series of terms, two per v_bitop3_b32: `acc |= a; acc |= b` is fused by the compiler
// into v_or3_b32 (4 issue cycles), the same truth table as a bitop3 issues in 2.  idx & 1 == 0: the term is parked in `pend`,
// == 1: acc = acc | pend | term; a series of odd length ORs the last parked term in itself.
__device__ __forceinline__ void or_pairs(uint32_t &acc, uint32_t &pend, int odd, uint32_t term) {
  if (odd) acc = B3(acc, pend, term, T_OR3);
  else pend = term;
}

// complete horizontal run fill of seeds s (subset of m), 6 ops: 2 carry fills, 2 bit reversals
__device__ __forceinline__ uint32_t run_fill2(uint32_t m, uint32_t mrev, uint32_t s) {
  uint32_t t = m + s;
  uint32_t u = B3(t, s, m, T_SEL);
  uint32_t rs = __brev(u);
  uint32_t t2 = mrev + rs;
  uint32_t rr = B3(t2, rs, mrev, T_SEL);
  return __brev(rr);
}
#define VISIT(r, nb) f[r] = run_fill2(m[r], mrev[r], B3(f[nb], m[r], f[r], T_ANDOR))

// Per-lane flood to the fixed point, variant for the per-ply kernels (next_states, children, 1-ply rollouts),
// which are stall-bound rather than issue-bound: two interleaved dependency chains per round for ILP (measured 15-25 % faster there than the serial
// schedule below, which in turn is 20 % faster in the fused rollout):
//   phase 1: chain A sweeps DOWN over the top rows [0..H], chain B sweeps UP over the bottom rows [R-1..H+1]
//   phase 2: chain B goes on UP over the top rows [H..0], chain A goes on DOWN over the bottom rows [H+1..R-1]
// After a round the top half is closed upwards, the bottom half downwards and the seam downwards; the test
// looks at the 18 remaining (row, direction) pairs and only then another round is spent.
template <int R>
__device__ __forceinline__ void flood2_dual(const uint32_t (&m)[R], const uint32_t (&mrev)[R], uint32_t (&f)[R],
                                            uint32_t *out) {
  constexpr int H = (R - 1) / 2;
#pragma unroll 1
  for (int it = 0; it < R * R; ++it) {
    f[0] = run_fill2(m[0], mrev[0], f[0]);
    f[R - 1] = run_fill2(m[R - 1], mrev[R - 1], f[R - 1]);
#pragma unroll
    for (int i = 1; i <= H; ++i) {
      VISIT(i, i - 1);
      if (R - 1 - i > H) VISIT(R - 1 - i, R - i);
    }
#pragma unroll
    for (int i = 0; i <= H; ++i) {
      VISIT(H - i, H - i + 1);
      if (H + 1 + i < R) VISIT(H + 1 + i, H + i);
    }
    uint32_t open = 0;
#pragma unroll
    for (int r = 1; r <= H; ++r) open |= B3(f[r - 1], m[r], f[r], T_AND_ANDN);      // top half, downwards
#pragma unroll
    for (int r = H; r < R - 1; ++r) open |= B3(f[r + 1], m[r], f[r], T_AND_ANDN);   // seam + bottom half, upwards
    if (__ballot(open != 0) == 0) break;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) out[r] = f[r];
}


// Flood variant for the fused rollout (issue-bound): whole-board Gauss-Seidel sweeps in one dependency chain, alternately DOWN and
// UP.  A sweep leaves the fill closed in its own direction (and horizontally - every visit is a complete run fill),
// so after a sweep only the opposite direction has to be tested: 18 three-input tests.  Measured on mid-game 19x19
// boards (all 44 floods of the wave must agree): down + up is almost never enough (0.2 %), down + up + down nearly
// always is (an arch-shaped group seeded at one foot needs exactly that), so the schedule is D, U, then
// {D, test, U, test}*; snake-shaped groups just take more sweeps (bounded by R*R).  The two-chain variant above needs
// one more sweep-equivalent on average: 1.41e9 vs 1.59e9 steps/s in the fused rollout.
// One v_bfrev per visit instead of two: the fill state alternates its BIT ORDER.  Before a down sweep row r is
// stored in domain (r & 1) (0 = normal, 1 = bit-reversed); a visit fills towards the MSB in the row's current
// domain, flips the row and fills towards the MSB again (i.e. the other board direction), leaving the row in the
// other domain - which is exactly the domain the next row (down sweep) / previous row (up sweep) is waiting in.
// All domains are compile-time constants of the unrolled code.  The closure test works on a normal-order copy of
// the rows (half of them need a v_bfrev); when it passes, that copy is the result.
// REV = the row is currently bit-reversed; NB = index of the neighbour row swept just before (or -1)
#define FLOOD_VISIT(r, NB, REV)                                                        \
  do {                                                                                 \
    const uint32_t ma_ = (REV) ? mrev[r] : m[r], mb_ = (REV) ? m[r] : mrev[r];         \
    const uint32_t s_ = ((NB) >= 0 && (NB) < R) ? B3(f[(NB) >= 0 && (NB) < R ? (NB) : 0], ma_, f[r], T_ANDOR) : f[r]; \
    const uint32_t t_ = ma_ + s_;                                                      \
    const uint32_t u_ = B3(t_, s_, ma_, T_SEL);                                        \
    const uint32_t v_ = __brev(u_);                                                    \
    const uint32_t t2_ = mb_ + v_;                                                     \
    f[r] = B3(t2_, v_, mb_, T_SEL);                                                    \
  } while (0)

// `out` = this lane's row of the L2 -> L1 transpose buffer: the converged fill is stored there in normal bit order
// (the normal-order copy made for the closure test is the result, so it never has to stay live across sweeps).
// PREREV: the caller hands the odd rows of the seeds over bit-reversed already
// OUT128: `out` is 16-byte aligned with room for (R + 3) & ~3 words; the copy is written four rows at a time
// EARLY: the first closure test already after the second sweep (down, up).  It pays on 9x9 boards in the from-scratch
// analysis (groups are small: config 2 fused 2.32 -> 2.38e9 steps/s, the 9x9 per-ply kernels -3 %) and nowhere else
// (13x13: -3 % in the fused rollout; 19x19: 2.33 -> 2.43 ms; the multi-ply kernel's floods at 9x9: -1.5 %)
#ifdef GG_AB_SWEEPS
// A/B builds only: sweeps per flood batch of the kernels that ask for it (gg_sweeps[0] sweeps, [1] batches)
static __device__ unsigned long long gg_sweeps[2];
#define GG_SWEEP_COUNT(n) do { if (COUNT) { int l_; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_)); \
    if (l_ == 0) { atomicAdd(&gg_sweeps[0], (unsigned long long)(n)); atomicAdd(&gg_sweeps[1], 1ull); } } } while (0)
#else
#define GG_SWEEP_COUNT(n) do {} while (0)
#endif
template <int R, bool PREREV = false, bool OUT128 = false, bool EARLY = false, bool COUNT = false>
__device__ __forceinline__ void flood2_serial(const uint32_t (&m)[R], const uint32_t (&mrev)[R], uint32_t (&f)[R],
                                              uint32_t *out) {
  if (!PREREV) {
#pragma unroll
    for (int r = 1; r < R; r += 2) f[r] = __brev(f[r]);  // seeds arrive in normal order
  }
#pragma unroll 1
  for (int it = 0; it < R * R; ++it) {
#pragma unroll
    for (int r = 0; r < R; ++r) FLOOD_VISIT(r, r - 1, (r & 1) != 0);       // down: domain (r&1) -> ((r+1)&1)
    if (it > 0) {
      // normal-order copy streamed into `out` (speculatively: it is the result if the test passes)
      uint32_t open = 0, pend = 0, above = 0;  // a filled stone whose upper neighbour is fillable but not filled
      uint32_t q[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int r = R - 1; r >= 0; --r) {
        const uint32_t g = ((r + 1) & 1) ? __brev(f[r]) : f[r];
        if (OUT128) {   // four rows per ds_write_b128: a lane's rows are RS (a multiple of 4) words apart, so 32-bit
          q[r & 3] = g; // stores of one row hit every fourth bank only (4-way conflict), 16-byte stores do not
          if ((r & 3) == 0) *reinterpret_cast<uint4 *>(out + r) = make_uint4(q[0], q[1], q[2], q[3]);
        } else {
          out[r] = g;
        }
        if (r < R - 1) or_pairs(open, pend, (R - 2 - r) & 1, B3(above, m[r], g, T_AND_ANDN));
        above = g;
      }
      if ((R - 1) & 1) open |= pend;
      if (__ballot(open != 0) == 0) { GG_SWEEP_COUNT(2 * it + 1); return; }
    }
#pragma unroll
    for (int r = R - 1; r >= 0; --r) FLOOD_VISIT(r, r + 1, ((r + 1) & 1) != 0);  // up: domain ((r+1)&1) -> (r&1)
    if (it > 0 || EARLY) {
      uint32_t open = 0, pend = 0, below = 0;
      uint32_t q[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t g = (r & 1) ? __brev(f[r]) : f[r];
        if (OUT128) {
          q[r & 3] = g;
          if ((r & 3) == 3 || r == R - 1) {
            if ((r & 3) != 3) { for (int z = (r & 3) + 1; z < 4; ++z) q[z] = 0u; }
            *reinterpret_cast<uint4 *>(out + (r & ~3)) = make_uint4(q[0], q[1], q[2], q[3]);
          }
        } else {
          out[r] = g;
        }
        if (r > 0) or_pairs(open, pend, (r - 1) & 1, B3(below, m[r], g, T_AND_ANDN));
        below = g;
      }
      if ((R - 1) & 1) open |= pend;
      if (__ballot(open != 0) == 0) { GG_SWEEP_COUNT(2 * it + 2); return; }
    }
  }
  // iteration bound hit (cannot happen for R <= 19): rows are in domain (r & 1)
#pragma unroll
  for (int r = 0; r < R; ++r) out[r] = (r & 1) ? __brev(f[r]) : f[r];
}


// Tromp-Taylor areas (gym_go/gogame.py:275-300) of sixteen boards whose stone rows sit in LDS as
// st[colour * 16 RS + board * RS + row] (rows >= N zero): a colour owns its stones plus the empty regions that touch only
// that colour, and an empty region touches a colour iff the flood of the EMPTY points seeded next to that colour's stones
// covers it.  Lane 2 s + c (the lower half of the wave) floods board s for colour c and returns that colour's area;
// `blocks` (32 RS words) receives the floods, `idle_out` (32 RS words) the rows of the idle upper half.
template <int R, bool FULLN>
__device__ __forceinline__ uint32_t areas16(const uint32_t *st, uint32_t *blocks, uint32_t *idle_out, int N, int lane) {
  constexpr int RS = Cfg<R>::kRowStride, RV = (R + 3) / 4, PL = 16 * RS;
  const int s = (lane >> 1) & 15, c = lane & 1;
  uint32_t cnt = 0;
  {
    uint32_t m[R], mrev[R], f[R];
    {
      uint32_t own[RV * 4 + 1], oth[RV * 4];
      const uint4 *po = reinterpret_cast<const uint4 *>(st + c * PL + s * RS);
      const uint4 *pt = reinterpret_cast<const uint4 *>(st + (1 - c) * PL + s * RS);
#pragma unroll
      for (int i = 0; i < RV; ++i) {
        const uint4 a = po[i], d = pt[i];
        own[4 * i] = a.x; own[4 * i + 1] = a.y; own[4 * i + 2] = a.z; own[4 * i + 3] = a.w;
        oth[4 * i] = d.x; oth[4 * i + 1] = d.y; oth[4 * i + 2] = d.z; oth[4 * i + 3] = d.w;
      }
      own[RV * 4] = 0;
      const uint32_t use = lane < 32 ? (1u << N) - 1u : 0u;   // the upper half of the wave carries no flood
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t e = (FULLN || r < N) ? B3(own[r], oth[r], use, ~(TA | TB) & TC & 0xFF) : 0u;   // empty points
        const uint32_t x = r > 0 ? B3(shl1(own[r]), own[r] >> 1, own[r - 1], T_OR3) : (shl1(own[r]) | (own[r] >> 1));
        m[r] = e;
        mrev[r] = __brev(e);
        f[r] = B3(e, x, r < R - 1 ? own[r + 1] : 0u, T_AND_OR2);
        cnt += (uint32_t)__popc(own[r]);
      }
    }
    WAVE_SYNC();
    flood2_serial<R>(m, mrev, f, lane < 32 ? blocks + lane * RS : idle_out + (lane - 32) * RS);
  }
  WAVE_SYNC();
  uint32_t fo[RV * 4], fp[RV * 4];
  const uint4 *pf = reinterpret_cast<const uint4 *>(blocks + (lane & 31) * RS);
  const uint4 *pp = reinterpret_cast<const uint4 *>(blocks + ((lane & 31) ^ 1) * RS);
#pragma unroll
  for (int i = 0; i < RV; ++i) {
    const uint4 a = pf[i], d = pp[i];
    fo[4 * i] = a.x; fo[4 * i + 1] = a.y; fo[4 * i + 2] = a.z; fo[4 * i + 3] = a.w;
    fp[4 * i] = d.x; fp[4 * i + 1] = d.y; fp[4 * i + 2] = d.z; fp[4 * i + 3] = d.w;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) cnt += (uint32_t)__popc(fo[r] & ~fp[r]);   // (word R of a block is never written)
  return cnt;
}


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

// ---------------------------------------------------------------- which pairs of boards a wave of a persistent grid takes
// A persistent per-pair kernel runs the resident set: `cols` SIMDs x W waves, and the dispatcher places workgroups c,
// c + cols, c + 2 cols ... on the same SIMD in that order (tools/ubench/placement.hip, tools/exp/where_ns.py (round 3, in git history)).  The arbiter
// of a SIMD serves its oldest wave first, so W waves with equal shares finish one after the other and the SIMD idles
// towards the end of the launch.  The pairs of a column (c, c + cols, c + 2 cols ...) are therefore split UNEVENLY by
// age: wave r of the column takes the pairs [cut[r-1], cut[r]) of it (cumulative 16.16 fractions; cut[-1] = 0, unused
// ranks hold 1.0, the youngest runs to the end).  cols == 0 (small batches, or a kernel whose occupancy is not known): pair p0 + k * grid.
constexpr int kAgeRanks = 4;   // waves per SIMD the split has (measured) shares for
struct AgeSplit {
  int cols;
  uint32_t cut[kAgeRanks - 1];
};
struct PairSpan {
  int64_t first, stride, end;
};
__device__ __forceinline__ PairSpan pair_span(int64_t npairs, const AgeSplit &as) {
  PairSpan sp;
  if (as.cols > 0) {
    const int c = (int)(blockIdx.x % (unsigned)as.cols), r = (int)(blockIdx.x / (unsigned)as.cols);
    const int64_t K = (npairs - c + as.cols - 1) / as.cols;   // pairs of this column
    const int64_t a0 = r == 0 ? 0 : (K * as.cut[r - 1 < kAgeRanks - 1 ? r - 1 : kAgeRanks - 2]) >> 16;
    const int64_t a1 = r >= kAgeRanks - 1 ? K : (K * as.cut[r]) >> 16;
    sp.stride = as.cols;
    sp.first = c + sp.stride * a0;
    sp.end = c + sp.stride * a1;
    if (sp.end > npairs) sp.end = npairs;
  } else {
    sp.first = blockIdx.x;
    sp.stride = gridDim.x;
    sp.end = npairs;
  }
  return sp;
}

// ---------------------------------------------------------------- a fair share of the SIMD for long-running waves
// The instruction arbiter of a SIMD serves its OLDEST wave first.  Four waves that each run 256 plies on one SIMD therefore
// do not advance together: the oldest runs at the speed of a wave that is alone (3.4 us per ply), the youngest gets what
// is left, and they finish at 0.40 / 0.59 / 0.79 / 1.00 of the launch - its second half runs at three, two and finally one
// wave per SIMD, where the issue port is idle most of the time (measured per workgroup with `s_getreg HW_ID` +
// wall_clock64: tools/exp/where.py).  Nothing a wave computes depends on another wave, so the cure is pure scheduling:
// every wave publishes how far it is (one word per hardware wave slot in a board in device memory), reads the words of
// the other slots of its SIMD, and sets its own issue priority (`s_setprio`, 0..3) by how many of its mates are at least
// `lag` units behind it - the leader yields, the stragglers catch up, all finish within microseconds of each other and
// the SIMD keeps four waves to the end: 2.33 -> 2.03 ms per 256-ply launch of the fused rollout (7.2 -> 8.3e9 steps/s).
// A hysteresis (`lag` = an eighth of the launch, at most 24 plies, checked every fourth ply) beats strict equality: the waves then hold DISTINCT priorities
// for long stretches, and four strictly ordered waves issue more per cycle than four that keep overtaking each other.
// The mates' words travel global -> LDS by LDS-DMA and are read at the NEXT check: no register is held and nothing
// waits for the load.  The board carries no result: stale or foreign entries (another stream's kernel) only shift
// priorities.  It is the one piece of mutable device-global state of the library (512 KB per device).
// (internal linkage: the library is two translation units - gg_kernels.hip and gg_rollout.hip - and each has its own board;
// waves of kernels from the other unit are "foreign entries" to it, like another process's)
static __device__ unsigned int gg_fair_board[8 * 8 * 2 * 16 * 4 * 16];   // [XCC][SE][SH][CU][SIMD][wave slot]: progress + 1, 0 / ~0 = free

__device__ __forceinline__ uint32_t lds_addr_of(const void *p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // M0 is written and consumed inside one asm statement (clang warns that it is reserved)
struct FairShare {
  unsigned int *row;      // this SIMD's 16 wave slots on the board (wave-uniform)
  unsigned int slot;      // this wave's slot
  uint32_t *mates;        // 16 words of LDS: the row as it was at the previous check
  uint32_t mates_lds;
  // active = false (a compile-time constant at the call): the launch never calls update / release, nothing is set up
  __device__ __forceinline__ explicit FairShare(uint32_t *lds16, bool active = true) : mates(lds16), mates_lds(lds_addr_of(lds16)) {
    row = nullptr;
    slot = 0;
    if (!active) return;
    unsigned int hw, xc;
    asm volatile("s_getreg_b32 %0, hwreg(4)" : "=s"(hw));    // HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]
    asm volatile("s_getreg_b32 %0, hwreg(20)" : "=s"(xc));   // XCC_ID
    const unsigned int simd = ((((xc & 7u) * 8u + ((hw >> 13) & 7u)) * 2u + ((hw >> 12) & 1u)) * 16u + ((hw >> 8) & 15u)) * 4u + ((hw >> 4) & 3u);
    row = gg_fair_board + simd * 16u;
    slot = hw & 15u;
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    if (ln < 16) lds16[ln] = 0u;   // nobody seen yet
  }
  // progress: any monotone counter (plies, iterations); lag: how far behind a mate must be to count
  __device__ __forceinline__ void update(uint32_t progress, uint32_t lag) {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the row requested at the previous check has landed long ago
    const uint32_t mate = ln < 16 ? mates[ln] : 0u;
    const uint32_t behind = (uint32_t)__popcll(__ballot(mate != 0u && mate != 0xFFFFFFFFu && mate + lag <= progress + 1u));
    if (behind >= 3u) __builtin_amdgcn_s_setprio(0);
    else if (behind == 2u) __builtin_amdgcn_s_setprio(1);
    else if (behind == 1u) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
    if (ln == 0) __hip_atomic_store(row + slot, progress + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS words are in registers before the DMA may overwrite them
    if (ln < 16) {   // next check's view of the row: global -> LDS directly (no VGPR is held while it flies)
      const unsigned int *src = row + ln;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off sc0 sc1" ::"s"(mates_lds), "v"(src) : "memory", "m0");
    }
  }
  // the same exchange without the priority: returns how many mates are >= lag units behind (wave-uniform); the caller sets
  // its priority itself (k_rollout5 combines it with the phase of the ply it is in)
  __device__ __forceinline__ uint32_t behind(uint32_t progress, uint32_t lag) {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t mate = ln < 16 ? mates[ln] : 0u;
    const uint32_t n = (uint32_t)__popcll(__ballot(mate != 0u && mate != 0xFFFFFFFFu && mate + lag <= progress + 1u));
    if (ln == 0) __hip_atomic_store(row + slot, progress + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (ln < 16) {
      const unsigned int *src = row + ln;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off sc0 sc1" ::"s"(mates_lds), "v"(src) : "memory", "m0");
    }
    return n;
  }
  __device__ __forceinline__ void release() {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    if (ln == 0) __hip_atomic_store(row + slot, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_setprio(0);
  }
};
#pragma clang diagnostic pop

// ---- sampler shared by the rollout kernels (mirrors oracle/gg_oracle.c splitmix_next / rollout_ply)
__device__ __forceinline__ uint64_t splitmix_next(uint64_t &x) {
  uint64_t z = (x += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// GoVecEnv auto-reset: games whose game-over plane is set are zeroed IN PLACE (build-side policy);
// one wave per finished board does the stores, everyone else only reads one byte.
static __global__ void k_reset_finished(uint8_t *__restrict__ states, int64_t B, int N) {
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

static __global__ void k_rng_seed(uint64_t *rng, uint64_t base_seed, int64_t first_game, int64_t B) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= B) return;
  uint64_t x = base_seed ^ ((uint64_t)(first_game + i) * 0xD1342543DE82EF95ull);
  splitmix_next(x);
  rng[i] = x;
}

}  // namespace gg
