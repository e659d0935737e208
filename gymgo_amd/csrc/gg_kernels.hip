// gg_kernels.hip - MI355X (gfx950 / CDNA4) kernels of the batched Go step path + the C-ABI.
//
// Design: ONE WAVEFRONT PER BOARD, 64-thread workgroups (so the workgroup barrier is a
// wave-local LDS fence and every wave runs its own data-dependent loops), persistent waves that
// grid-stride over the batch.  Integer / bit work only - no MFMA.
//
// A board lives in two register layouts:
//   L1 "row per lane":   lane r (< N) holds row r of a plane as a 32-bit mask (bit c = column c).
//                        Point-wise rules are 1 VALU op for the whole board; vertical neighbours are
//                        one cross-lane move.
//   L2 "flood per lane": every lane holds ALL rows of one colour (R registers) plus its own fill.
//                        Each lane runs a DIFFERENT flood fill of the same board at the same time:
//                        lanes 0-19 next mover's groups, lanes 32-51 mover's groups, one lane per
//                        liberty class (bit k of the row / column index == v, k < 5, v in {0,1}).
//                        A flood = Gauss-Seidel row sweeps (down, up) with a carry-propagate
//                        horizontal run fill, so a sweep crosses the whole board.
// From the 40 floods: a group reached from both classes (k,0) and (k,1) for some k has >= 2 distinct
// liberties; a group reached by neither class of k = 0 has none (captured); everything else has
// exactly one.  That is all the reference's invalid-move rule needs (state_utils.py:24-83 restated
// point-wise): an empty point is playable iff a neighbour is empty, or a next-mover
// stone with >= 2 liberties, or a mover stone with exactly 1.
//
// Reference citations are path:line relative to the reference root (huangeddie/GymGo).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gymgo_amd.h"

namespace {

constexpr int kWave = 64;
constexpr int kClassBits = 5;                 // row / column indices < 32
constexpr int kClasses = 4 * kClassBits;      // (row|col bit k) x (value v) = 20 floods per colour

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

// Fill every maximal run of `m` that contains a bit of `s` (s subset of m), both directions.
// Up-fill: t = m + s carries from each seed to the end of its run; (t&s)|(~t&m) keeps exactly the
// bits from the lowest seed of a run upwards.  Down-fill = the same in the bit-reversed domain.
__device__ __forceinline__ uint32_t run_fill(uint32_t m, uint32_t mrev, uint32_t s) {
  uint32_t t = m + s;
  uint32_t u = (t & s) | (~t & m);
  uint32_t rs = __brev(u);
  uint32_t t2 = mrev + rs;
  uint32_t rr = (t2 & rs) | (~t2 & mrev);
  return __brev(rr);
}

// Per-lane flood fill of `f` (seeds) through mask `m` to the fixed point.  All 64 lanes run their own
// flood in lock-step.  A down sweep leaves f closed horizontally and downwards, an up sweep
// horizontally and upwards; after each sweep a 2-op-per-row test asks whether any lane could still
// grow in the opposite direction, and only then is another sweep spent.
template <int R>
__device__ __forceinline__ void flood(const uint32_t (&m)[R], const uint32_t (&mrev)[R], uint32_t (&f)[R]) {
  f[0] = run_fill(m[0], mrev[0], f[0]);
#pragma unroll
  for (int r = 1; r < R; ++r) f[r] = run_fill(m[r], mrev[r], f[r] | (f[r - 1] & m[r]));
#pragma unroll 1
  for (int it = 0; it < R * R; ++it) {
#pragma unroll
    for (int r = R - 2; r >= 0; --r) f[r] = run_fill(m[r], mrev[r], f[r] | (f[r + 1] & m[r]));
    uint32_t open_dn = 0;  // a filled stone whose lower neighbour is fillable but not filled
#pragma unroll
    for (int r = 1; r < R; ++r) open_dn |= f[r - 1] & m[r] & ~f[r];
    if (__ballot(open_dn != 0) == 0) break;
#pragma unroll
    for (int r = 1; r < R; ++r) f[r] = run_fill(m[r], mrev[r], f[r] | (f[r - 1] & m[r]));
    uint32_t open_up = 0;
#pragma unroll
    for (int r = 0; r < R - 1; ++r) open_up |= f[r + 1] & m[r] & ~f[r];
    if (__ballot(open_up != 0) == 0) break;
  }
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
  flood<R>(mm, mr, f);
#pragma unroll
  for (int r = 0; r < R; ++r) sc[lane * RS + r] = f[r];
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

// ---------------------------------------------------------------- staging: HBM <-> LDS <-> bitboards
// Boards start at arbitrary byte offsets (6 N^2 is only a multiple of 2), so every wide access below is
// an UNALIGNED 16- or 4-byte access; gfx950 runs with unaligned global and LDS access enabled and the
// compiler emits global_load/store_dwordx4, ds_read_b128, ds_write_b32 for these packed types.
struct __attribute__((packed, aligned(1))) V16u { uint32_t w[4]; };
struct __attribute__((packed, aligned(1))) W32u { uint32_t v; };
struct __attribute__((aligned(16))) V16a { uint32_t w[4]; };

__device__ __forceinline__ uint32_t ld32u(const uint8_t *p) { return reinterpret_cast<const W32u *>(p)->v; }
__device__ __forceinline__ void st32u(uint8_t *p, uint32_t v) { reinterpret_cast<W32u *>(p)->v = v; }

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

// row / column of a flat action with the host-supplied reciprocal: inv = ceil(2^16 / N), exact for a <= N*N
__device__ __forceinline__ void split_action(int a, int N, uint32_t inv, int &r, int &c) {
  r = (int)(((uint32_t)a * inv) >> 16);
  c = a - r * N;
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
    flood<R>(m, mrev, f);
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

// ===================================================================== v2: TWO BOARDS PER WAVEFRONT
// Lanes 0-31 own board A, lanes 32-63 board B (h = lane >> 5, hl = lane & 31).  Everything that is
// wave-uniform in v1 (action, turn, pass / done flags, ko point) is a per-lane value that is equal
// inside a half; ballots are split into their 32-bit halves.
//
// Liberty classes: instead of 20 (bit, value) classes, a CONSTANT-WEIGHT CODE - point q = 19 r + c gets
// the q-th 11-bit word of weight 5 (C(11,5) = 462 >= 361); flood i (11 per colour, 22 lanes per board) is
// seeded from the empty points whose word has bit i.  A group with one liberty is reached by exactly 5
// floods, a group with two or more distinct liberties by >= 6 (two different weight-5 words), a group with
// none by 0: a bit-sliced population count over the 11 floods (carry-save adders, ~20 L1 ops) classifies
// every stone of the board at once.
//
// Instruction selection (tools/ubench/valu_rate2.hip, measured on MI355X): v_and/or/xor/add/sub/lshrrev/
// bitop3/mov issue in 2 cycles per wave64; v_bfrev, v_and_or, v_or3, v_lshl_or, v_lshlrev, v_bfi, v_bcnt,
// v_bfe, v_mul_u32_u24, v_dot4, v_readlane cost 4.  The hot loops below therefore spell every 3-input
// boolean as v_bitop3_b32 and every "<< 1" as an add.
constexpr int kCwClasses = 11, kCwWeight = 5, kCwLanes = 2 * kCwClasses;

struct CwTable { uint32_t m[kCwClasses + 1][20]; };  // [class][row] -> columns of the class; last row = zeros

constexpr CwTable make_cw_table() {
  CwTable t{};
  int q = 0;
  for (uint32_t w = 0; w < (1u << kCwClasses) && q < 19 * 19; ++w) {
    int pc = 0;
    for (int i = 0; i < kCwClasses; ++i) pc += (w >> i) & 1u;
    if (pc != kCwWeight) continue;
    const int r = q / 19, c = q % 19;
    for (int i = 0; i < kCwClasses; ++i)
      if ((w >> i) & 1u) t.m[i][r] |= 1u << c;
    ++q;
  }
  return t;
}
__constant__ CwTable kCw = make_cw_table();

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

struct Half {
  int lane, h, hl;
  int N, P;
  uint32_t inv, full_l1;
  int cls;       // flood class of this lane (kCwClasses = idle lane)
  bool second;   // lane floods the second colour
};

__device__ __forceinline__ uint32_t half_of(uint64_t ballot, int h) {
  return h ? (uint32_t)(ballot >> 32) : (uint32_t)ballot;
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

// Per-lane flood to the fixed point, two interleaved dependency chains per round for ILP:
//   phase 1: chain A sweeps DOWN over the top rows [0..H], chain B sweeps UP over the bottom rows [R-1..H+1]
//   phase 2: chain B goes on UP over the top rows [H..0], chain A goes on DOWN over the bottom rows [H+1..R-1]
// After a round the top half is closed upwards, the bottom half downwards and the seam downwards; the test
// looks at the 18 remaining (row, direction) pairs and only then another round is spent.
template <int R>
__device__ __forceinline__ void flood2(const uint32_t (&m)[R], const uint32_t (&mrev)[R], uint32_t (&f)[R]) {
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
}

// sum bit (a^b^c) and carry bit (majority) of a bit-sliced full adder: one v_bitop3_b32 each
__device__ __forceinline__ uint32_t csa_sum(uint32_t a, uint32_t b, uint32_t c) { return B3(a, b, c, T_XOR3); }
__device__ __forceinline__ uint32_t csa_carry(uint32_t a, uint32_t b, uint32_t c) { return B3(a, b, c, T_MAJ); }

// From the 11 floods of one colour (w[i] = this row's bits reached by flood i): alive = reached by any,
// multi = reached by >= 6.
__device__ __forceinline__ void classify11(const uint32_t (&w)[kCwClasses], uint32_t &alive, uint32_t &multi) {
  uint32_t s0 = csa_sum(w[0], w[1], w[2]), c0 = csa_carry(w[0], w[1], w[2]);
  uint32_t s1 = csa_sum(w[3], w[4], w[5]), c1 = csa_carry(w[3], w[4], w[5]);
  uint32_t s2 = csa_sum(w[6], w[7], w[8]), c2 = csa_carry(w[6], w[7], w[8]);
  uint32_t s3 = w[9] ^ w[10], c3 = w[9] & w[10];
  uint32_t ss = csa_sum(s0, s1, s2), cs = csa_carry(s0, s1, s2);
  uint32_t t = ss & s3;                      // ones column done (bit 0 itself is not needed)
  uint32_t u0 = csa_sum(c0, c1, c2), v0 = csa_carry(c0, c1, c2);
  uint32_t u1 = csa_sum(c3, cs, t), v1 = csa_carry(c3, cs, t);
  uint32_t bit1 = u0 ^ u1, v2 = u0 & u1;
  uint32_t bit2 = csa_sum(v0, v1, v2), bit3 = csa_carry(v0, v1, v2);
  multi = B3(bit3, bit2, bit1, T_OR_AND);                 // count >= 6
  alive = B3(ss, s3, bit1, T_OR3) | bit2 | bit3;          // count >= 1
}

// LDS carve-up of a v2 workgroup: the flood transpose buffer and the board staging buffers are never live
// at the same time and share region 0.
template <int R>
struct Lds2 {
  static constexpr int kScWords = kWave * Cfg<R>::kRowStride;
  static constexpr int kIoWords = 2 * Cfg<R>::kIoBytes / 4;
  static constexpr int kRegion0 = kScWords > kIoWords ? kScWords : kIoWords;
  static constexpr int kRows5 = kRegion0;                 // [2][160]
  static constexpr int kCwt = kRows5 + 2 * 160;           // [12][20]
  static constexpr int kTbl = kCwt + (kCwClasses + 1) * 20;  // [16] 4 bits -> 4 bytes
  static constexpr int kTotal = kTbl + 16;
};

// Liberty analysis of both boards of the wave (L1 in, L1 out; see analyze<R> for the single-board form).
template <int R>
__device__ __forceinline__ void analyze2(uint32_t c0, uint32_t c1, uint32_t e, const Half &hf, uint32_t *lds,
                                         uint32_t &multi0, uint32_t &alive0, uint32_t &multi1) {
  constexpr int RS = Cfg<R>::kRowStride;
  constexpr int RV = (R + 3) / 4;
  uint32_t *sc = lds;
  uint32_t *my5 = lds + Lds2<R>::kRows5 + hf.h * 160;
  const uint32_t *cwt = lds + Lds2<R>::kCwt;
  WAVE_SYNC();
  my5[hf.hl] = c0;
  my5[32 + hf.hl] = c1;
  my5[64 + hf.hl] = __brev(c0);
  my5[96 + hf.hl] = __brev(c1);
  my5[128 + hf.hl] = e;
  WAVE_SYNC();
  uint32_t m[R], mrev[R], f[R];
  {
    uint32_t ee[RV * 4 + 1], mt[RV * 4];
    const uint4 *pm = reinterpret_cast<const uint4 *>(my5 + (hf.second ? 32 : 0));
    const uint4 *pe = reinterpret_cast<const uint4 *>(my5 + 128);
    const uint4 *pc = reinterpret_cast<const uint4 *>(cwt + hf.cls * 20);
#pragma unroll
    for (int i = 0; i < RV; ++i) {
      uint4 a = pm[i], c = pe[i], d = pc[i];
      mt[4 * i] = a.x; mt[4 * i + 1] = a.y; mt[4 * i + 2] = a.z; mt[4 * i + 3] = a.w;
      ee[4 * i] = c.x & d.x; ee[4 * i + 1] = c.y & d.y; ee[4 * i + 2] = c.z & d.z; ee[4 * i + 3] = c.w & d.w;
    }
    ee[RV * 4] = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      m[r] = mt[r];
      // stones touching a liberty of the class: m & ((e << 1) | (e >> 1) | e_above | e_below)
      uint32_t x = r > 0 ? B3(shl1(ee[r]), ee[r] >> 1, ee[r - 1], T_OR3) : (shl1(ee[r]) | (ee[r] >> 1));
      f[r] = B3(m[r], x, r < R - 1 ? ee[r + 1] : 0u, T_AND_OR2);
    }
    const uint4 *pr = reinterpret_cast<const uint4 *>(my5 + 64 + (hf.second ? 32 : 0));
#pragma unroll
    for (int i = 0; i < RV; ++i) {
      uint4 b = pr[i];
      if (4 * i < R) mrev[4 * i] = b.x;
      if (4 * i + 1 < R) mrev[4 * i + 1] = b.y;
      if (4 * i + 2 < R) mrev[4 * i + 2] = b.z;
      if (4 * i + 3 < R) mrev[4 * i + 3] = b.w;
    }
  }
  flood2<R>(m, mrev, f);
#pragma unroll
  for (int r = 0; r < R; ++r) sc[hf.lane * RS + r] = f[r];
  WAVE_SYNC();
  multi0 = 0; multi1 = 0; alive0 = 0;
  if (hf.hl < R) {
    const uint32_t *base = sc + (hf.h * 32) * RS + hf.hl;
    uint32_t w0[kCwClasses], w1[kCwClasses];
#pragma unroll
    for (int i = 0; i < kCwClasses; ++i) {
      w0[i] = base[i * RS];
      w1[i] = base[(kCwClasses + i) * RS];
    }
    uint32_t alive1;
    classify11(w0, alive0, multi0);
    classify11(w1, alive1, multi1);
  }
}

__device__ __forceinline__ uint32_t invalid_from2(uint32_t nx, uint32_t pl, uint32_t multi_nx, uint32_t multi_pl,
                                                  const Half &hf) {
  uint32_t e = hf.full_l1 & ~(nx | pl);
  uint32_t x = B3(e, nx & multi_nx, pl & ~multi_pl, T_OR3);
  // rows above / below: one-lane DPP shifts over the whole wave; rows >= N are zero, so nothing leaks
  // across the half boundary (N <= 19 < 32)
  uint32_t up = dpp0<0x138>(x), dn = dpp0<0x130>(x);
  uint32_t nb = B3(shl1(x), x >> 1, up, T_OR3) | dn;
  return hf.full_l1 & ~(e & nb);
}

// One transition per half (see step_core<R>).  `a` is this half's action (a legal point or P).
// atari_in (valid when have_atari, which must be wave-uniform) = the opponent's stones whose group had exactly one
// liberty BEFORE the move, as classified by the previous ply's analysis: a group of that set touching the new
// stone loses its last liberty, so the captures are known up front (a few L1 flood steps through the atari set)
// and ONE analysis of the final position suffices.  Without it the first analysis finds the liberty-less groups and
// a second one re-analyses (~21 % of wave passes).  atari_out = the mover's stones in atari after the move.
template <int R>
__device__ __forceinline__ uint32_t step_core2(uint32_t &mine, uint32_t &opp, int a, const Half &hf, uint32_t *lds,
                                               uint32_t atari_in, bool have_atari, uint32_t &atari_out) {
  const bool is_pass = a >= hf.P;
  int ko_r = -1, ko_c = 0;
  bool boxed = false;
  uint32_t nbm = 0;
  {
    const int aa = is_pass ? 0 : a;
    const int ra = (int)(((uint32_t)aa * hf.inv) >> 16), ca = aa - ra * hf.N;
    const uint32_t bit = is_pass ? 0u : (1u << ca);
    if (hf.hl == ra) mine |= bit;
    if (hf.hl == ra) nbm = (bit << 1) | (bit >> 1);
    if (hf.hl == ra - 1 || hf.hl == ra + 1) nbm = bit;
    nbm &= hf.full_l1;
    boxed = half_of(__ballot((nbm & ~opp) != 0), hf.h) == 0;
  }
  // gogame.py:72-75 - remove `dead`, ko iff exactly one stone died and the new stone is boxed in
  auto capture = [&](uint32_t dead) {
    uint32_t dm = half_of(__ballot(dead != 0), hf.h);
    uint32_t many = half_of(__ballot(__popc(dead) > 1), hf.h);
    int r = dm ? (__ffs(dm) - 1) : 0;
    uint32_t drow = __shfl(dead, (hf.lane & 32) + r);
    if (dm && boxed && many == 0 && (dm & (dm - 1)) == 0) {
      ko_r = r;
      ko_c = __ffs(drow) - 1;
    }
    opp &= ~dead;
  };
  uint32_t multi_opp, alive_opp, multi_mine;
  if (have_atari) {
    uint32_t f = nbm & atari_in;  // atari groups touching the new stone ...
    if (__ballot(f != 0)) {
#pragma unroll 1
      for (int it = 0; it < R * R; ++it) {  // ... completed through the atari set
        uint32_t grow = B3(shl1(f), f >> 1, dpp0<0x138>(f), T_OR3) | dpp0<0x130>(f);
        uint32_t g = B3(grow, atari_in, f, T_ANDOR);
        const bool ch = g != f;
        f = g;
        if (__ballot(ch) == 0) break;
      }
      capture(f);
    }
    uint32_t e = hf.full_l1 & ~(mine | opp);
    analyze2<R>(opp, mine, e, hf, lds, multi_opp, alive_opp, multi_mine);
  } else {
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      uint32_t e = hf.full_l1 & ~(mine | opp);
      analyze2<R>(opp, mine, e, hf, lds, multi_opp, alive_opp, multi_mine);
      if (pass == 0) {
        uint32_t dead = is_pass ? 0u : (opp & ~alive_opp);
        if (__ballot(dead != 0)) {  // some board of the wave captured: fix it up, analyse both again
          capture(dead);
          continue;
        }
      }
      break;
    }
  }
  atari_out = mine & ~multi_mine;
  uint32_t invalid = invalid_from2(opp, mine, multi_opp, multi_mine, hf);
  if (hf.hl == ko_r) invalid |= 1u << ko_c;
  return invalid;
}

// per-half staging: the 32 lanes of a half move their own board.
// Boards start at arbitrary byte offsets, but unaligned 16-byte global accesses run at about half the rate of
// aligned ones on gfx950 (tools/time_align.py: I/O overhead 70-80 us per 65 536-board launch vs 42 us for the
// 16-byte-aligned N = 16 stride).  So HBM is only ever touched with ALIGNED 16-byte vectors: the load fetches the
// aligned superset of the slice (the extra <= 30 bytes belong to neighbouring boards or to the same 16-byte
// chunk as the first / last valid byte, hence to a mapped page) and the board lives at offset mis = g & 15
// inside the LDS buffer; the store writes the fully covered aligned vectors and ONE global_store_byte
// instruction whose lanes 0-14 / 16-30 carry the ragged head / tail bytes.
__device__ __forceinline__ uint32_t stage_in_h(const uint8_t *g, int nbytes, uint8_t *lds, int hl) {
  const uint32_t mis = (uint32_t)((uintptr_t)g & 15u);
  const uint8_t *ga = g - mis;
  const int nv = (int)(mis + nbytes + 15) >> 4;
  for (int v = hl; v < nv; v += 32)
    *reinterpret_cast<V16a *>(lds + 16 * v) = *reinterpret_cast<const V16a *>(ga + 16 * v);
  return mis;
}

// lds[mis + j] = board byte j, mis = g & 15
__device__ __forceinline__ void stage_out_h(uint8_t *g, int nbytes, const uint8_t *lds, int hl, bool on) {
  if (!on) return;
  const uint32_t mis = (uint32_t)((uintptr_t)g & 15u);
  uint8_t *ga = g - mis;
  const int end = (int)mis + nbytes;
  const int v0 = mis ? 1 : 0, v1 = end >> 4;
  for (int v = v0 + hl; v < v1; v += 32)
    *reinterpret_cast<V16a *>(ga + 16 * v) = *reinterpret_cast<const V16a *>(lds + 16 * v);
  if (v1 >= v0) {
    const int head = mis ? 16 - (int)mis : 0, tail = end & 15;
    int j = -1;
    if (hl < 16) { if (hl < head) j = hl; }
    else if (hl - 16 < tail) j = nbytes - tail + (hl - 16);
    if (j >= 0) g[j] = lds[mis + j];
  } else {  // the slice lies inside one 16-byte chunk
    for (int i = hl; i < nbytes; i += 32) g[i] = lds[mis + i];
  }
}

// rare path: an illegal move's row passes through unchanged, global -> global, bytes
__device__ __forceinline__ void copy_row_h(const uint8_t *src, uint8_t *dst, int nbytes, int hl, bool on) {
  if (!on) return;
  for (int i = hl; i < nbytes; i += 32) dst[i] = src[i];
}

// Board emission of one half, L1 rows -> HBM, with ALIGNED LDS and HBM accesses only:
//   1. the 6 planes are OR-ed row by row (ds_or_b32) into a linear bit-string bs[] (bit 16 + i = board byte i;
//      the 16 leading zero bits stand for the bytes in front of the board inside its first 16-byte chunk);
//   2. lane v of round k builds the aligned 16-byte vector 16 (hl + 32 k): 16 cells = one funnel shift out of
//      two words of bs[], 4 cells -> 4 bytes through a 16-entry table (aligned ds_read_b32);
//   3. vectors that lie inside the board go straight from registers to HBM (global_store_dwordx4); the (at
//      most two) ragged ones are parked in LDS and leave in ONE global_store_byte instruction.
// `work` = the half's LDS staging area (>= 96 + 8 words), `tbl` = the bits -> bytes table.
template <int R>
__device__ __forceinline__ void emit_store_h(uint8_t *g, uint32_t black, uint32_t white, uint32_t invalid,
                                             uint32_t turn, uint32_t passed, uint32_t done, const Half &hf,
                                             uint32_t *work, const uint32_t *tbl, bool wr) {
  constexpr int kRounds = (Cfg<R>::kIoBytes / 16 + 31) / 32;
  uint32_t *bs = work;
  uint8_t *edge = reinterpret_cast<uint8_t *>(work + 96);  // [2][16]
  const int S = 6 * hf.P;
  WAVE_SYNC();
  bs[hf.hl] = 0; bs[32 + hf.hl] = 0; bs[64 + hf.hl] = 0;
  WAVE_SYNC();
  if (wr && hf.hl < hf.N) {
    const uint32_t rows[6] = {black, white, turn ? hf.full_l1 : 0u, invalid, passed ? hf.full_l1 : 0u,
                              done ? hf.full_l1 : 0u};
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const uint32_t q = 16u + (uint32_t)(p * hf.P + hf.hl * hf.N), w = q >> 5, sh = q & 31u;
      if (rows[p]) {
        atomicOr(&bs[w], rows[p] << sh);
        if (sh + (uint32_t)hf.N > 32u) atomicOr(&bs[w + 1], rows[p] >> (32u - sh));
      }
    }
  }
  WAVE_SYNC();
  const uint32_t mo = (uint32_t)((uintptr_t)g & 15u);
  uint8_t *ga = g - mo;
  const int nv = (int)(mo + S + 15) >> 4;
  if (wr) {
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
      const int v = hf.hl + 32 * k;
      if (v < nv) {
        const uint32_t qb = 16u + 16u * (uint32_t)v - mo, w = qb >> 5, sh = qb & 31u;
        const uint32_t b16 = __builtin_amdgcn_alignbit(bs[w + 1], bs[w], sh);
        V16a o;
        o.w[0] = tbl[b16 & 15u];
        o.w[1] = tbl[(b16 >> 4) & 15u];
        o.w[2] = tbl[(b16 >> 8) & 15u];
        o.w[3] = tbl[(b16 >> 12) & 15u];
        const int lo = 16 * v - (int)mo;
        const bool full = lo >= 0 && lo + 16 <= S;
        if (full) *reinterpret_cast<V16a *>(ga + 16 * v) = o;   // HBM, aligned
        asm volatile("" ::: "memory");                          // keep the two address spaces apart (no flat store)
        if (!full) {
          uint32_t *e = work + 96 + (lo < 0 ? 0 : 4);
          e[0] = o.w[0]; e[1] = o.w[1]; e[2] = o.w[2]; e[3] = o.w[3];
        }
      }
    }
  }
  WAVE_SYNC();
  if (wr) {
    const int head = mo ? 16 - (int)mo : 0, tail = ((int)mo + S) & 15;
    if (nv >= 2) {
      int j = -1, e = 0;
      if (hf.hl < 16) { if (hf.hl < head) { j = hf.hl; e = (int)mo + hf.hl; } }
      else if (hf.hl - 16 < tail) { j = S - tail + (hf.hl - 16); e = 16 + (hf.hl - 16); }
      if (j >= 0) g[j] = edge[e];
    } else {  // the whole board sits in one 16-byte chunk (N = 2 with a lucky offset never happens: S >= 24)
      for (int i = hf.hl; i < S; i += 32) g[i] = edge[(mo ? 0 : 16) + ((int)mo + i)];
    }
  }
}

__device__ __forceinline__ uint32_t load_flags_h(const uint8_t *g, int P, int pt, const Half &hf) {
  uint8_t fb = 0;
  if (hf.hl < 4) {
    int off = hf.hl == 0 ? 2 * P : hf.hl == 1 ? 3 * P + pt : hf.hl == 2 ? 4 * P : 5 * P;
    fb = g[off];
  }
  return half_of(__ballot(fb != 0), hf.h) & 0xFu;
}

__device__ __forceinline__ Half make_half(int lane, int N, uint32_t inv) {
  Half hf;
  hf.lane = lane; hf.h = lane >> 5; hf.hl = lane & 31;
  hf.N = N; hf.P = N * N; hf.inv = inv;
  hf.full_l1 = hf.hl < N ? (1u << N) - 1u : 0u;
  hf.cls = hf.hl < kCwLanes ? (hf.hl % kCwClasses) : kCwClasses;
  hf.second = hf.hl >= kCwClasses;
  return hf;
}

template <int R>
__device__ __forceinline__ void load_cw_table(uint32_t *lds, int lane) {
  uint32_t *cwt = lds + Lds2<R>::kCwt;
  for (int i = lane; i < (kCwClasses + 1) * 20; i += kWave) cwt[i] = kCw.m[i / 20][i % 20];
  if (lane < 16)  // bits -> bytes expansion table of emit_store_h
    lds[Lds2<R>::kTbl + lane] = (lane & 1u) | ((lane & 2u) << 7) | ((lane & 4u) << 14) | ((lane & 8u) << 21);
  WAVE_SYNC();
}

// inclusive prefix sum of v over the 32 lanes of each half: 4 DPP row shifts + 1 row broadcast
__device__ __forceinline__ uint32_t half_scan(uint32_t v) {
  v += dpp0<0x111>(v);
  v += dpp0<0x112>(v);
  v += dpp0<0x114>(v);
  v += dpp0<0x118>(v);
  v += dpp0<0x142, 0xA>(v);  // lane 15 of rows 0 / 2 added to every lane of rows 1 / 3
  return v;
}

// k-th valid action of this half's board (see pick_action); incl = half_scan(popc(valid))
__device__ __forceinline__ int pick_action2(uint32_t valid, uint32_t incl, uint32_t k, const Half &hf) {
  uint32_t hit = half_of(__ballot(incl > k), hf.h);
  int r = hit ? (__ffs(hit) - 1) : 0;
  int src = (hf.lane & 32) + r;
  uint32_t row = __shfl(valid, src);
  uint32_t before = (uint32_t)__shfl((int)incl, src) - (uint32_t)__popc(row);
  uint32_t t = k - before;
  bool me = ((row >> hf.hl) & 1u) && (uint32_t)__popc(row & ((1u << hf.hl) - 1u)) == t;
  uint32_t cb = half_of(__ballot(me), hf.h);
  int c = cb ? (__ffs(cb) - 1) : 0;
  return hit ? r * hf.N + c : hf.P;
}

template <int R>
__global__ __launch_bounds__(kWave, 4) void k_next_states2(const uint8_t *__restrict__ in,
                                                        const int32_t *__restrict__ actions,
                                                        uint8_t *__restrict__ out, int32_t *__restrict__ status,
                                                        int64_t B, int N, uint32_t inv, int canonical) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[Lds2<R>::kTotal];
  const Half hf = make_half(threadIdx.x, N, inv);
  load_cw_table<R>(lds, hf.lane);
  const int S = 6 * hf.P;
  uint8_t *io = reinterpret_cast<uint8_t *>(lds) + hf.h * Cfg<R>::kIoBytes;
  const int64_t npairs = (B + 1) >> 1;
  for (int64_t p = blockIdx.x; p < npairs; p += gridDim.x) {
    const int64_t b0 = 2 * p + hf.h;
    const bool on = b0 < B;
    const int64_t b = on ? b0 : B - 1;
    const uint8_t *gi = in + b * (int64_t)S;
    uint8_t *go = out + b * (int64_t)S;
    int a = actions[b];
    const bool in_range = a >= 0 && a <= hf.P;
    const bool is_pass = a == hf.P;
    uint32_t flags = load_flags_h(gi, hf.P, (in_range && !is_pass) ? a : 0, hf);
    const bool illegal = !in_range || (!is_pass && (flags & 2u));
    WAVE_SYNC();
    const uint32_t mi = stage_in_h(gi, 2 * hf.P, io, hf.hl);
    WAVE_SYNC();
    if (__ballot(!illegal) == 0) {  // both rows pass through unchanged (gogame.py:59 / :117 would raise)
      copy_row_h(gi, go, S, hf.hl, on);
      if (status && on && hf.hl == 0) status[b] = GG_STATUS_ILLEGAL;
      continue;
    }
    uint32_t black = plane_to_row<R>(io + mi, N, hf.hl);
    uint32_t white = plane_to_row<R>(io + mi + hf.P, N, hf.hl);
    const int pl = flags & 1u;
    uint32_t mine = pl ? white : black, opp = pl ? black : white;
    // an illegal half still runs the (wave-wide) analysis on a harmless pass, its result is discarded
    uint32_t atari_unused;
    uint32_t invalid = step_core2<R>(mine, opp, illegal ? hf.P : a, hf, lds, 0u, false, atari_unused);
    black = pl ? opp : mine;
    white = pl ? mine : opp;
    uint32_t passed = is_pass ? 1 : 0;
    uint32_t done = ((flags & 8u) || (is_pass && (flags & 4u))) ? 1 : 0;
    int nturn = 1 - pl;
    if (canonical && nturn == 1) {
      uint32_t t = black; black = white; white = t;
      nturn = 0;
    }
    emit_store_h<R>(go, black, white, invalid, (uint32_t)nturn, passed, done, hf,
                    reinterpret_cast<uint32_t *>(io), lds + Lds2<R>::kTbl, on && !illegal);
    if (illegal) copy_row_h(gi, go, S, hf.hl, on);  // rare: the row passes through unchanged
    if (status && on && hf.hl == 0) status[b] = illegal ? GG_STATUS_ILLEGAL : GG_STATUS_OK;
  }
}

template <int R>
__global__ __launch_bounds__(kWave, 4) void k_rollout2(uint8_t *__restrict__ states, uint64_t *__restrict__ rng,
                                                    int32_t *__restrict__ last_actions,
                                                    int64_t *__restrict__ steps_done, int64_t B, int N, uint32_t inv,
                                                    int plies, int auto_reset) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[Lds2<R>::kTotal];
  const Half hf = make_half(threadIdx.x, N, inv);
  load_cw_table<R>(lds, hf.lane);
  const int S = 6 * hf.P;
  uint8_t *io = reinterpret_cast<uint8_t *>(lds) + hf.h * Cfg<R>::kIoBytes;
  const int64_t npairs = (B + 1) >> 1;
  for (int64_t p = blockIdx.x; p < npairs; p += gridDim.x) {
    const int64_t bA = 2 * p, bB = (2 * p + 1 < B) ? 2 * p + 1 : B - 1;
    const bool on = 2 * p + hf.h < B;
    const int64_t b = hf.h ? bB : bA;
    uint8_t *gs = states + b * (int64_t)S;
    uint32_t flags = load_flags_h(gs, hf.P, 0, hf);
    WAVE_SYNC();
    const uint32_t mi = stage_in_h(gs, 4 * hf.P, io, hf.hl);
    WAVE_SYNC();
    uint32_t black = plane_to_row<R>(io + mi, N, hf.hl);
    uint32_t white = plane_to_row<R>(io + mi + hf.P, N, hf.hl);
    uint32_t invalid = plane_to_row<R>(io + mi + 3 * hf.P, N, hf.hl);
    int turn = flags & 1u, passed = (flags >> 2) & 1u, done = (flags >> 3) & 1u;
    uint64_t xa = uniform64(rng[bA]), xb = uniform64(rng[bB]);  // generator states live in SGPRs
    int last = -1, played = 0;
    uint32_t atari = 0;   // next mover's opponents in atari, known from the previous ply of this launch
    bool have_atari = false;
#pragma unroll 1
    for (int t = 0; t < plies; ++t) {
      const bool live = on && !(done && !auto_reset);
      const uint64_t lv = __ballot(live);
      if (lv == 0) break;
      if (done && live) {
        black = white = invalid = 0;
        turn = passed = done = 0;
        atari = 0;  // empty board: nothing is in atari
      }
      uint32_t valid = hf.full_l1 & ~invalid;
      uint32_t incl = half_scan((uint32_t)__popc(valid));
      uint32_t cnt_a = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31);
      uint32_t cnt_b = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      uint64_t xna = xa, xnb = xb;
      uint64_t ua = splitmix_next(xna), ub = splitmix_next(xnb);
      uint32_t ka = (uint32_t)(((ua >> 32) * (uint64_t)(cnt_a + 1)) >> 32);
      uint32_t kb = (uint32_t)(((ub >> 32) * (uint64_t)(cnt_b + 1)) >> 32);
      if ((uint32_t)lv) xa = xna;
      if ((uint32_t)(lv >> 32)) xb = xnb;
      int a = pick_action2(valid, incl, hf.h ? kb : ka, hf);
      uint32_t mine = turn ? white : black, opp = turn ? black : white;
      uint32_t natari;
      uint32_t ninv = step_core2<R>(mine, opp, live ? a : hf.P, hf, lds, atari, have_atari, natari);
      have_atari = true;   // from now on every live half carries its atari set (frozen halves only ever pass)
      if (live) {
        atari = natari;
        invalid = ninv;
        black = turn ? opp : mine;
        white = turn ? mine : opp;
        if (a == hf.P) { if (passed) done = 1; passed = 1; } else passed = 0;
        turn ^= 1;
        last = a;
        ++played;
      }
    }
    if (__ballot(played != 0)) {
      emit_store_h<R>(gs, black, white, invalid, (uint32_t)turn, (uint32_t)passed, (uint32_t)done, hf,
                      reinterpret_cast<uint32_t *>(io), lds + Lds2<R>::kTbl, on && played != 0);
    }
    if (on && hf.hl == 0) {
      rng[b] = hf.h ? xb : xa;
      if (last_actions) last_actions[b] = last;
      if (steps_done) steps_done[b] += played;
    }
  }
}

// all-zero child slot straight from registers (no LDS round trip), aligned vectors + one byte-store for the edges
__device__ __forceinline__ void stage_zero_h(uint8_t *g, int nbytes, int hl, bool on) {
  if (!on) return;
  const uint32_t mis = (uint32_t)((uintptr_t)g & 15u);
  uint8_t *ga = g - mis;
  const int end = (int)mis + nbytes;
  const int v0 = mis ? 1 : 0, v1 = end >> 4;
  const V16a z = {{0u, 0u, 0u, 0u}};
  for (int v = v0 + hl; v < v1; v += 32) *reinterpret_cast<V16a *>(ga + 16 * v) = z;
  if (v1 >= v0) {
    const int head = mis ? 16 - (int)mis : 0, tail = end & 15;
    int j = -1;
    if (hl < 16) { if (hl < head) j = hl; }
    else if (hl - 16 < tail) j = nbytes - tail + (hl - 16);
    if (j >= 0) g[j] = 0;
  } else {
    for (int i = hl; i < nbytes; i += 32) g[i] = 0;
  }
}

// gogame.children, two slots per wave pass.  The legal actions of the chunk are compacted first (k-th set bit of the
// valid-point rows, as in the sampler) so that both halves always expand a legal action; the all-zero slots of
// the illegal actions are written in a separate store-only loop.
template <int R>
__global__ __launch_bounds__(kWave, 4) void k_children2(const uint8_t *__restrict__ states,
                                                        uint8_t *__restrict__ children, int64_t B, int N,
                                                        uint32_t inv, int canonical, int chunks) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[Lds2<R>::kTotal];
  const Half hf = make_half(threadIdx.x, N, inv);
  load_cw_table<R>(lds, hf.lane);
  const int S = 6 * hf.P;
  const int A = hf.P + 1;
  uint8_t *io = reinterpret_cast<uint8_t *>(lds) + hf.h * Cfg<R>::kIoBytes;
  const int per = (A + chunks - 1) / chunks;
  for (int64_t w = blockIdx.x; w < B * chunks; w += gridDim.x) {
    const int64_t b = w / chunks;
    const int ch = (int)(w - b * chunks);
    const uint8_t *gi = states + b * (int64_t)S;
    uint8_t *gc = children + b * A * (int64_t)S;
    uint32_t flags = load_flags_h(gi, hf.P, 0, hf);
    WAVE_SYNC();
    const uint32_t mi = stage_in_h(gi, 4 * hf.P, io, hf.hl);  // both halves stage the same parent (second copy: L2)
    WAVE_SYNC();
    const uint32_t black = plane_to_row<R>(io + mi, N, hf.hl);
    const uint32_t white = plane_to_row<R>(io + mi + hf.P, N, hf.hl);
    const uint32_t invd = plane_to_row<R>(io + mi + 3 * hf.P, N, hf.hl);
    const int pl = flags & 1u;
    const int a0 = ch * per, a1 = min(A, a0 + per);
    const int p1 = min(a1, hf.P);  // points of the chunk: [a0, p1); the pass slot is in the chunk iff a1 == A
    // rows of this chunk's points
    const int base = hf.hl * N;
    const int lo = max(0, min(N, a0 - base)), hi = max(0, min(N, p1 - base));
    const uint32_t inrange = (hi > lo) ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
    const uint32_t vr = hf.full_l1 & ~invd & inrange;
    const uint32_t incl = half_scan((uint32_t)__popc(vr));
    const int npts = __builtin_amdgcn_readlane((int)incl, 31);
    const int nv = npts + (a1 == A ? 1 : 0);
    // The all-zero slots of the illegal points are store-only work; they are interleaved with the compute passes
    // (q zero steps after every pass) so that the write stream is spread over the whole life of the wave.
    int az = a0;
    const int npass = (nv + 1) >> 1, nzero = (p1 - a0 + 1) >> 1;
    const int q = npass > 0 ? (nzero + npass - 1) / npass : nzero;
    auto zero_step = [&](int aj) {
      const int a = aj + hf.h;
      bool zero = false;
      if (a < p1) {
        const int ra = (int)(((uint32_t)a * inv) >> 16), ca = a - ra * N;
        uint32_t row = __shfl(invd, (hf.lane & 32) + ra);
        zero = ((row >> ca) & 1u) != 0;
      }
      stage_zero_h(gc + (int64_t)(a < p1 ? a : a0) * S, S, hf.hl, zero);
    };
#pragma unroll 1
    for (int j = 0; j < nv; j += 2) {
      const int k = j + hf.h;
      const bool on = k < nv;
      const int a = on ? pick_action2(vr, incl, (uint32_t)k, hf) : hf.P;  // k == npts -> pass
      const bool is_pass = a == hf.P;
      uint32_t mine = pl ? white : black, opp = pl ? black : white;
      uint32_t atari_unused;
      uint32_t invalid = step_core2<R>(mine, opp, a, hf, lds, 0u, false, atari_unused);
      uint32_t nb = pl ? opp : mine, nw = pl ? mine : opp;
      uint32_t passed = is_pass ? 1 : 0;
      uint32_t done = ((flags & 8u) || (is_pass && (flags & 4u))) ? 1 : 0;
      int nturn = 1 - pl;
      if (canonical && nturn == 1) {
        uint32_t t = nb; nb = nw; nw = t;
        nturn = 0;
      }
      uint8_t *go = gc + (int64_t)a * S;
      emit_store_h<R>(go, nb, nw, invalid, (uint32_t)nturn, passed, done, hf, reinterpret_cast<uint32_t *>(io),
                      lds + Lds2<R>::kTbl, on);
      for (int t = 0; t < q && az < p1; ++t, az += 2) zero_step(az);
    }
    for (; az < p1; az += 2) zero_step(az);
  }
}

// ---------------------------------------------------------------- host side

int g_cus = -1;

// kernel family: 2 = two boards per wavefront (default), 1 = one board per wavefront.  GG_KERNEL_VARIANT=1|2.
int variant() {
  static int v = 0;
  if (v == 0) {
    const char *e = getenv("GG_KERNEL_VARIANT");
    v = (e && e[0] == '1') ? 1 : 2;
  }
  return v;
}

int device_cus() {
  if (g_cus < 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    g_cus = cus;
  }
  return g_cus;
}

// persistent grid: enough single-wave workgroups to fill every SIMD several times over
int grid_for(int64_t work) {
  int cus = device_cus();
  if (cus <= 0) cus = 256;
  int64_t cap = (int64_t)cus * 32;
  return (int)(work < cap ? (work > 0 ? work : 1) : cap);
}

int32_t check(int64_t B, int32_t N) { return (N < 2 || N > GG_MAX_BOARD || B < 0) ? GG_E_BADSIZE : 0; }

// reciprocal for the in-kernel action -> (row, col) split; exactness is verified for every action
uint32_t recip16(int32_t N) {
  uint32_t inv = (65536u + (uint32_t)N - 1u) / (uint32_t)N;
  for (uint32_t a = 0; a <= (uint32_t)(N * N); ++a)
    if (((a * inv) >> 16) != a / (uint32_t)N) return 0;
  return inv;
}

#define GG_DISPATCH(N, CALL9, CALL13, CALL19) \
  do {                                        \
    if ((N) <= 9) { CALL9; }                  \
    else if ((N) <= 13) { CALL13; }           \
    else { CALL19; }                          \
  } while (0)

}  // namespace

extern "C" {

int32_t gg_version(void) { return GG_ABI_VERSION; }

int32_t gg_device_cus(void) { return device_cus(); }

int32_t gg_batch_next_states(const uint8_t *in, const int32_t *actions, uint8_t *out, int32_t *status, int64_t B,
                             int32_t N, int32_t canonical, void *hip_stream) {
  if (int32_t e = check(B, N)) return e;
  if (B == 0) return 0;
  if (!in || !actions || !out) return GG_E_NULLPTR;
  const uint32_t inv = recip16(N);
  if (!inv) return GG_E_BADSIZE;
  hipStream_t s = (hipStream_t)hip_stream;
  int grid = grid_for(B);
  if (variant() == 2) {
    grid = grid_for((B + 1) / 2);
    GG_DISPATCH(N, (k_next_states2<9><<<grid, kWave, 0, s>>>(in, actions, out, status, B, N, inv, canonical)),
                (k_next_states2<13><<<grid, kWave, 0, s>>>(in, actions, out, status, B, N, inv, canonical)),
                (k_next_states2<19><<<grid, kWave, 0, s>>>(in, actions, out, status, B, N, inv, canonical)));
  } else {
    GG_DISPATCH(N, (k_next_states<9><<<grid, kWave, 0, s>>>(in, actions, out, status, B, N, inv, canonical)),
                (k_next_states<13><<<grid, kWave, 0, s>>>(in, actions, out, status, B, N, inv, canonical)),
                (k_next_states<19><<<grid, kWave, 0, s>>>(in, actions, out, status, B, N, inv, canonical)));
  }
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_invalid_mask(const uint8_t *states, const int32_t *ko, uint8_t *mask, int64_t B, int32_t N,
                              void *hip_stream) {
  if (int32_t e = check(B, N)) return e;
  if (B == 0) return 0;
  if (!states || !mask) return GG_E_NULLPTR;
  const uint32_t inv = recip16(N);
  if (!inv) return GG_E_BADSIZE;
  hipStream_t s = (hipStream_t)hip_stream;
  int grid = grid_for(B);
  GG_DISPATCH(N, (k_invalid_mask<9><<<grid, kWave, 0, s>>>(states, ko, mask, B, N, inv)),
              (k_invalid_mask<13><<<grid, kWave, 0, s>>>(states, ko, mask, B, N, inv)),
              (k_invalid_mask<19><<<grid, kWave, 0, s>>>(states, ko, mask, B, N, inv)));
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_areas(const uint8_t *states, int32_t *black, int32_t *white, int64_t B, int32_t N, void *hip_stream) {
  if (int32_t e = check(B, N)) return e;
  if (B == 0) return 0;
  if (!states || !black || !white) return GG_E_NULLPTR;
  hipStream_t s = (hipStream_t)hip_stream;
  int grid = grid_for(B);
  GG_DISPATCH(N, (k_areas<9><<<grid, kWave, 0, s>>>(states, black, white, B, N)),
              (k_areas<13><<<grid, kWave, 0, s>>>(states, black, white, B, N)),
              (k_areas<19><<<grid, kWave, 0, s>>>(states, black, white, B, N)));
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_children(const uint8_t *states, uint8_t *children, int64_t B, int32_t N, int32_t canonical,
                          void *hip_stream) {
  if (int32_t e = check(B, N)) return e;
  if (B == 0) return 0;
  if (!states || !children) return GG_E_NULLPTR;
  const uint32_t inv = recip16(N);
  if (!inv) return GG_E_BADSIZE;
  hipStream_t s = (hipStream_t)hip_stream;
  const int A = N * N + 1;
  int cus = device_cus();
  if (cus <= 0) cus = 256;
  int64_t want = (int64_t)cus * 64;  // work items wanted to keep every SIMD busy
  int chunks = (int)((want + B - 1) / B);
  if (chunks < 1) chunks = 1;
  if (chunks > A) chunks = A;
  int grid = grid_for(B * chunks);
  if (variant() == 2) {
    GG_DISPATCH(N, (k_children2<9><<<grid, kWave, 0, s>>>(states, children, B, N, inv, canonical, chunks)),
                (k_children2<13><<<grid, kWave, 0, s>>>(states, children, B, N, inv, canonical, chunks)),
                (k_children2<19><<<grid, kWave, 0, s>>>(states, children, B, N, inv, canonical, chunks)));
  } else {
    GG_DISPATCH(N, (k_children<9><<<grid, kWave, 0, s>>>(states, children, B, N, inv, canonical, chunks)),
                (k_children<13><<<grid, kWave, 0, s>>>(states, children, B, N, inv, canonical, chunks)),
                (k_children<19><<<grid, kWave, 0, s>>>(states, children, B, N, inv, canonical, chunks)));
  }
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_rollout(uint8_t *states, uint64_t *rng, int32_t *last_actions, int64_t *steps_done, int64_t B,
                         int32_t N, int32_t plies, int32_t auto_reset, void *hip_stream) {
  if (int32_t e = check(B, N)) return e;
  if (plies < 0) return GG_E_BADARG;
  if (B == 0 || plies == 0) return 0;
  if (!states || !rng) return GG_E_NULLPTR;
  const uint32_t inv = recip16(N);
  if (!inv) return GG_E_BADSIZE;
  hipStream_t s = (hipStream_t)hip_stream;
  int grid = grid_for(B);
  if (variant() == 2) {
    grid = grid_for((B + 1) / 2);
    GG_DISPATCH(N, (k_rollout2<9><<<grid, kWave, 0, s>>>(states, rng, last_actions, steps_done, B, N, inv, plies, auto_reset)),
                (k_rollout2<13><<<grid, kWave, 0, s>>>(states, rng, last_actions, steps_done, B, N, inv, plies, auto_reset)),
                (k_rollout2<19><<<grid, kWave, 0, s>>>(states, rng, last_actions, steps_done, B, N, inv, plies, auto_reset)));
  } else {
    GG_DISPATCH(N, (k_rollout<9><<<grid, kWave, 0, s>>>(states, rng, last_actions, steps_done, B, N, inv, plies, auto_reset)),
                (k_rollout<13><<<grid, kWave, 0, s>>>(states, rng, last_actions, steps_done, B, N, inv, plies, auto_reset)),
                (k_rollout<19><<<grid, kWave, 0, s>>>(states, rng, last_actions, steps_done, B, N, inv, plies, auto_reset)));
  }
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_sample_actions(const uint8_t *states, uint64_t *rng, int32_t *actions, int64_t B, int32_t N,
                                void *hip_stream) {
  if (int32_t e = check(B, N)) return e;
  if (B == 0) return 0;
  if (!states || !rng || !actions) return GG_E_NULLPTR;
  hipStream_t s = (hipStream_t)hip_stream;
  int grid = grid_for(B);
  GG_DISPATCH(N, (k_sample<9><<<grid, kWave, 0, s>>>(states, rng, actions, B, N)),
              (k_sample<13><<<grid, kWave, 0, s>>>(states, rng, actions, B, N)),
              (k_sample<19><<<grid, kWave, 0, s>>>(states, rng, actions, B, N)));
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_update_pieces(uint8_t *states, const int32_t *points, const int32_t *players, uint8_t *killed,
                               int64_t B, int32_t N, void *hip_stream) {
  if (int32_t e = check(B, N)) return e;
  if (B == 0) return 0;
  if (!states || !points || !players) return GG_E_NULLPTR;
  const uint32_t inv = recip16(N);
  if (!inv) return GG_E_BADSIZE;
  hipStream_t s = (hipStream_t)hip_stream;
  int grid = grid_for(B);
  GG_DISPATCH(N, (k_update_pieces<9><<<grid, kWave, 0, s>>>(states, points, players, killed, B, N, inv)),
              (k_update_pieces<13><<<grid, kWave, 0, s>>>(states, points, players, killed, B, N, inv)),
              (k_update_pieces<19><<<grid, kWave, 0, s>>>(states, points, players, killed, B, N, inv)));
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_reset_finished(uint8_t *states, int64_t B, int32_t N, void *hip_stream) {
  if (int32_t e = check(B, N)) return e;
  if (B == 0) return 0;
  if (!states) return GG_E_NULLPTR;
  hipStream_t s = (hipStream_t)hip_stream;
  int64_t waves = (B + kWave - 1) / kWave;
  int blocks = (int)((waves + 3) / 4);
  if (blocks > 4096) blocks = 4096;
  k_reset_finished<<<blocks, 4 * kWave, 0, s>>>(states, B, N);
  return (int32_t)hipGetLastError();
}

int32_t gg_rng_seed(uint64_t *rng, uint64_t base_seed, int64_t first_game, int64_t B, void *hip_stream) {
  if (B < 0) return GG_E_BADSIZE;
  if (B == 0) return 0;
  if (!rng) return GG_E_NULLPTR;
  hipStream_t s = (hipStream_t)hip_stream;
  k_rng_seed<<<(unsigned)((B + 255) / 256), 256, 0, s>>>(rng, base_seed, first_game, B);
  return (int32_t)hipGetLastError();
}

}  // extern "C"
