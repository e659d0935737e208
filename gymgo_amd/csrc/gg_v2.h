// gg_v2.h - the per-ply kernels: TWO BOARDS PER WAVEFRONT, every liberty class from scratch (constant-weight code).
#pragma once
#include "gg_common.h"

namespace gg {

// ===================================================================== v2: TWO BOARDS PER WAVEFRONT
// Lanes 0-31 own board A, lanes 32-63 board B (h = lane >> 5, hl = lane & 31).  Everything that would be
// wave-uniform with one board per wave (action, turn, pass / done flags, ko point) is a per-lane value that is
// equal inside a half; ballots are split into their 32-bit halves.
//
// Liberty classes: a CONSTANT-WEIGHT CODE - point q = 19 r + c gets
// the q-th 11-bit word of weight 5 (C(11,5) = 462 >= 361); flood i (11 per colour, 22 lanes per board) is
// seeded from the empty points whose word has bit i.  A group with one liberty is reached by exactly 5
// floods, a group with two or more distinct liberties by >= 6 (two different weight-5 words), a group with
// none by 0: a bit-sliced population count over the 11 floods (carry-save adders, ~20 L1 ops) classifies
// every stone of the board at once.
//
// The flood variants, the 2-cycle op spelling (bitop3 / add-for-shift) and the staging helpers are in gg_common.h.
constexpr int kCwClasses = 11, kCwWeight = 5, kCwLanes = 2 * kCwClasses;

#ifdef GG_AB_WHERE
// A/B builds only: per workgroup (XCC, HW_ID, duration in 100 MHz ticks) - how evenly do the waves of a launch finish?
static __device__ unsigned int gg_where[3 * 16384];
#endif

#ifdef GG_AB_PROF
// A/B builds only: shader-clock time of the phases of a ply as one wave experiences them (incl. waiting for the SIMD), one
// record per (single-wave) workgroup - plain stores into its own slots, nothing shared; [8] / [9] = the workgroup's entry /
// exit on the 100 MHz wall clock (of its latest launch).  GG_PROF_READ reduces them on the host: sums, earliest entry, latest exit.
constexpr int kProfSlots = 16384;
static __device__ unsigned long long gg_prof[kProfSlots * 10];
#define GG_PROF_DECL unsigned long long tph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tw_ = (unsigned long long)wall_clock64(), tc_ = clock64()
#define GG_PROF(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); \
    const unsigned long long n_ = clock64(); __builtin_amdgcn_sched_barrier(0); tph_[k] += n_ - tc_; tc_ = n_; } while (0)
#define GG_PROF_FLUSH do { const unsigned int ws_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); \
    if ((threadIdx.x & 63) == 0 && ws_ < kProfSlots) { unsigned long long *o_ = gg_prof + 10 * ws_; \
    for (int k_ = 0; k_ < 8; ++k_) o_[k_] += tph_[k_]; \
    o_[8] = tw_; o_[9] = (unsigned long long)wall_clock64(); \
    for (int k_ = 0; k_ < 8; ++k_) tph_[k_] = 0; } } while (0)
#define GG_PROF_READ(name) extern "C" int32_t name(unsigned long long *out10) { \
    if (hipDeviceSynchronize() != hipSuccess) return 1; \
    static unsigned long long h_[gg::kProfSlots * 10]; \
    if (hipMemcpyFromSymbol(h_, HIP_SYMBOL(gg::gg_prof), sizeof(h_)) != hipSuccess) return 2; \
    for (int k = 0; k < 10; ++k) out10[k] = k == 8 ? ~0ull : 0; \
    for (int i = 0; i < gg::kProfSlots; ++i) { const unsigned long long *r = h_ + 10 * i; if (!r[9]) continue; \
      for (int k = 0; k < 8; ++k) out10[k] += r[k]; \
      if (r[8] < out10[8]) out10[8] = r[8]; if (r[9] > out10[9]) out10[9] = r[9]; } \
    for (int i = 0; i < gg::kProfSlots * 10; ++i) h_[i] = 0; \
    return hipMemcpyToSymbol(HIP_SYMBOL(gg::gg_prof), h_, sizeof(h_)) == hipSuccess ? 0 : 3; }
#define GG_PROF_RAW(name) extern "C" int32_t name(unsigned long long *out, int slots) { \
    if (hipDeviceSynchronize() != hipSuccess) return 1; \
    if (slots > gg::kProfSlots) slots = gg::kProfSlots; \
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(gg::gg_prof), sizeof(unsigned long long) * 10 * (size_t)slots) == hipSuccess ? 0 : 2; }
#elif defined(GG_AB_MARK)
// A/B builds only: phase markers in the assembly listing (tools/isa_mix.py --phases)
#define GG_PROF_DECL do {} while (0)
#define GG_PROF(k) asm volatile("; GGMARK " #k ::: "memory")
#define GG_PROF_FLUSH do {} while (0)
#else
#define GG_PROF_DECL do {} while (0)
#define GG_PROF(k) do {} while (0)
#define GG_PROF_FLUSH do {} while (0)
#endif

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
static __constant__ CwTable kCw = make_cw_table();   // (internal linkage: one copy per translation unit)

#ifndef GG_LB_CH3
#define GG_LB_CH3 4   // waves per SIMD k_children3 is compiled for (128 VGPRs, no spills; 3 waves / 140 VGPRs measures the same +-3 %)
#endif
#ifndef GG_LB_PLY
#define GG_LB_PLY 3   // waves per SIMD the per-ply kernels are compiled for
#endif

struct Half {
  int lane, h, hl;
  int N, P;
  uint32_t inv, full_l1;
  int cls;       // row of the class-mask table: 0..10 liberty class, 11 = all zeros (idle lane), 12 = all ones
  // word offsets of this lane's flood inputs inside its half's 6-plane row buffer (planes of 32 words:
  // 0 c0, 1 c1, 2 c0 reversed, 3 c1 reversed, 4 empties, 5 empties reversed)
  int m_off, r_off, s_off;
};

constexpr int kRowPlanes = 6, kRowBuf = kRowPlanes * 32;

__device__ __forceinline__ uint32_t half_of(uint64_t ballot, int h) {
  return h ? (uint32_t)(ballot >> 32) : (uint32_t)ballot;
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
  static constexpr int kRows5 = kRegion0;                 // [2][kRowBuf]
  static constexpr int kCwt = kRows5 + 2 * kRowBuf;       // [13][20]: 11 classes, zeros, ones
  static constexpr int kTotal = kCwt + (kCwClasses + 2) * 20;
};

// Liberty analysis of both boards of the wave (L1 in, L1 out; see analyze<R> for the single-board form).
// AREAS: also return reach0 / reach1 = the empty points connected (through empty points) to a neighbour of a c0 / c1
// stone (needs make_half(..., areas = true)).
template <int R, bool DUAL, bool AREAS = false>
__device__ __forceinline__ void analyze2(uint32_t c0, uint32_t c1, uint32_t e, const Half &hf, uint32_t *lds,
                                         uint32_t &multi0, uint32_t &alive0, uint32_t &multi1, uint32_t *reach = nullptr,
                                         uint32_t *alive1_out = nullptr, bool compact = false) {
  constexpr int RS = Cfg<R>::kRowStride;
  constexpr int RV = (R + 3) / 4;
  uint32_t *sc = lds;
  // compact (LDS-tight callers): the row planes alias the transpose buffer - they are read into registers before the
  // flood and the flood's first store to the buffer comes sweeps later (DS operations of a wave execute in order) -
  // and the class table is read from constant memory instead of its LDS copy: the scratch is region 0 alone.
  uint32_t *my5 = compact ? lds + hf.h * kRowBuf : lds + Lds2<R>::kRows5 + hf.h * kRowBuf;
  const uint32_t *cwt = compact ? &kCw.m[0][0] : lds + Lds2<R>::kCwt;
  WAVE_SYNC();
  my5[hf.hl] = c0;
  my5[32 + hf.hl] = c1;
  my5[64 + hf.hl] = __brev(c0);
  my5[96 + hf.hl] = __brev(c1);
  my5[128 + hf.hl] = e;
  if (AREAS) my5[160 + hf.hl] = __brev(e);
  WAVE_SYNC();
  uint32_t m[R], mrev[R], f[R];
  {
    uint32_t ee[RV * 4 + 1], mt[RV * 4];
    const uint4 *pm = reinterpret_cast<const uint4 *>(my5 + hf.m_off);
    const uint4 *pe = reinterpret_cast<const uint4 *>(my5 + hf.s_off);
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
      // points of the flood mask touching a seed source of the lane: m & ((s << 1) | (s >> 1) | s_above | s_below)
      uint32_t x = r > 0 ? B3(shl1(ee[r]), ee[r] >> 1, ee[r - 1], T_OR3) : (shl1(ee[r]) | (ee[r] >> 1));
      f[r] = B3(m[r], x, r < R - 1 ? ee[r + 1] : 0u, T_AND_OR2);
    }
    const uint4 *pr = reinterpret_cast<const uint4 *>(my5 + hf.r_off);
#pragma unroll
    for (int i = 0; i < RV; ++i) {
      uint4 b = pr[i];
      if (4 * i < R) mrev[4 * i] = b.x;
      if (4 * i + 1 < R) mrev[4 * i + 1] = b.y;
      if (4 * i + 2 < R) mrev[4 * i + 2] = b.z;
      if (4 * i + 3 < R) mrev[4 * i + 3] = b.w;
    }
  }
  if (DUAL) flood2_dual<R>(m, mrev, f, sc + hf.lane * RS);
  else flood2_serial<R, false, false, (R <= 9)>(m, mrev, f, sc + hf.lane * RS);
  WAVE_SYNC();
  multi0 = 0; multi1 = 0; alive0 = 0;
  if (alive1_out) *alive1_out = 0;
  if (AREAS) { reach[0] = 0; reach[1] = 0; }
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
    if (alive1_out) *alive1_out = alive1;
    if (AREAS) {
      reach[0] = base[kCwLanes * RS];
      reach[1] = base[(kCwLanes + 1) * RS];
    }
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
template <int R, bool DUAL, bool AREAS = false>
__device__ __forceinline__ uint32_t step_core2(uint32_t &mine, uint32_t &opp, int a, const Half &hf, uint32_t *lds,
                                               uint32_t atari_in, bool have_atari, uint32_t &atari_out,
                                               uint32_t *reach = nullptr) {
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
    analyze2<R, DUAL, AREAS>(opp, mine, e, hf, lds, multi_opp, alive_opp, multi_mine, reach);
  } else {
    uint32_t e = hf.full_l1 & ~(mine | opp);
    uint32_t alive_mine;
    analyze2<R, DUAL, AREAS>(opp, mine, e, hf, lds, multi_opp, alive_opp, multi_mine, reach, &alive_mine);
    const uint32_t dead = is_pass ? 0u : (opp & ~alive_opp);
    if (__ballot(dead != 0)) {   // some board of the wave captured
      capture(dead);
      if (AREAS) {
        // the territory floods saw the captured stones: analyse the final position again
        e = hf.full_l1 & ~(mine | opp);
        analyze2<R, DUAL, AREAS>(opp, mine, e, hf, lds, multi_opp, alive_opp, multi_mine, reach);
      } else {
        // No second analysis: removing `dead` only gives liberties to the mover's groups next to it.
        //  * the group of the new stone, if it had none (G0 = every mover's stone without a liberty): its liberties
        //    are exactly the captured points next to it;
        //  * a group in atari next to a captured stone now has >= 2 (its old liberty was an empty point, the new ones
        //    were stones) - completed by an L1 flood through the atari set;  groups with >= 2 keep >= 2.
        // The opponent's surviving groups touch no captured point (they would be the same group).
        const uint32_t G0 = mine & ~alive_mine;
        const uint32_t dG = B3(shl1(G0), G0 >> 1, dpp0<0x138>(G0), T_OR3) | dpp0<0x130>(G0);
        const uint32_t l0 = dG & dead;
        const uint32_t nz = half_of(__ballot(l0 != 0), hf.h), many = half_of(__ballot(__popc(l0) > 1), hf.h);
        const bool multi0 = ((nz & (nz - 1u)) | many) != 0;
        const uint32_t atari_m = mine & alive_mine & ~multi_mine;
        const uint32_t dd = B3(shl1(dead), dead >> 1, dpp0<0x138>(dead), T_OR3) | dpp0<0x130>(dead);
        uint32_t f = dd & atari_m;
        if (__ballot(f != 0)) {
#pragma unroll 1
          for (int it = 0; it < R * R; ++it) {
            const uint32_t grow = B3(shl1(f), f >> 1, dpp0<0x138>(f), T_OR3) | dpp0<0x130>(f);
            const uint32_t g = B3(grow, atari_m, f, T_ANDOR);
            const bool ch = g != f;
            f = g;
            if (__ballot(ch) == 0) break;
          }
        }
        multi_mine |= f | (multi0 ? G0 : 0u);
      }
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

// 8 cells (bits) -> 8 bytes, as a 256-entry table in LDS (the board emitters are VALU-bound: one ds_read_b64
// replaces two bit-field extracts, two 24-bit multiplies and two ANDs)
__device__ __forceinline__ void load_spread_lut(uint2 *lut, int lane) {
  for (int e = lane; e < 256; e += kWave)
    lut[e] = make_uint2(__umul24((uint32_t)e & 15u, 0x204081u) & 0x01010101u,
                        __umul24((uint32_t)e >> 4, 0x204081u) & 0x01010101u);
  WAVE_SYNC();
}

// Board emission of one half, L1 rows -> HBM, with ALIGNED LDS and HBM accesses only:
//   1. the 6 planes are OR-ed row by row (ds_or_b32) into a linear bit-string bs[]: bit (g & 15) + i = board
//      byte i, so that every aligned 16-byte vector of HBM is exactly one 16-bit halfword of bs[];
//   2. lane v of round k emits the aligned vector 16 (hl + 32 k): two bytes of bs[] index the spread table, the
//      two 8-byte entries go straight from registers to HBM (global_store_dwordx4);
//   3. the ragged head / tail bytes (vectors shared with the neighbouring boards) leave in ONE global_store_byte
//      instruction, each lane picking its bit out of bs[].
// `work` = the half's LDS scratch (>= 96 words), `lut` = load_spread_lut's table.
template <int R>
__device__ __forceinline__ void emit_store_h(uint8_t *g, uint32_t black, uint32_t white, uint32_t invalid,
                                             uint32_t turn, uint32_t passed, uint32_t done, const Half &hf,
                                             uint32_t *work, const uint2 *lut, bool wr) {
  constexpr int kRounds = (Cfg<R>::kIoBytes / 16 + 31) / 32;
  constexpr int kZero = ((15 + 6 * R * R + 31) / 32 + 31) / 32;
  static_assert(kZero <= 3, "work area is 96 words");
  uint32_t *bs = work;
  const int S = 6 * hf.P;
  const uint32_t mo = (uint32_t)((uintptr_t)g & 15u);
  WAVE_SYNC();
#pragma unroll
  for (int k = 0; k < kZero; ++k) bs[hf.hl + 32 * k] = 0;
  WAVE_SYNC();
  if (wr && hf.hl < hf.N) {
    const uint32_t rows[6] = {black, white, turn ? hf.full_l1 : 0u, invalid, passed ? hf.full_l1 : 0u,
                              done ? hf.full_l1 : 0u};
    const uint32_t q0 = mo + (uint32_t)(hf.hl * hf.N);
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      if (rows[p]) {
        const uint32_t q = q0 + (uint32_t)(p * hf.P);
        const uint64_t x = (uint64_t)rows[p] << (q & 31u);
        uint32_t *w = bs + (q >> 5);
        atomicOr(w, (uint32_t)x);
        if ((uint32_t)(x >> 32)) atomicOr(w + 1, (uint32_t)(x >> 32));
      }
    }
  }
  WAVE_SYNC();
  if (wr) {
    uint8_t *ga = g - mo;
    const int end = (int)mo + S;
    const int v0 = mo ? 1 : 0, v1 = end >> 4;
    const uint8_t *bb = reinterpret_cast<const uint8_t *>(bs);
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
      const int v = hf.hl + 32 * k;
      if (v >= v0 && v < v1) {
        const uint2 lo = lut[bb[2 * v]], hi = lut[bb[2 * v + 1]];
        V16a o;
        o.w[0] = lo.x; o.w[1] = lo.y; o.w[2] = hi.x; o.w[3] = hi.y;
        *reinterpret_cast<V16a *>(ga + 16 * v) = o;
      }
    }
    const int head = mo ? 16 - (int)mo : 0, tail = end & 15;
    int j = -1;
    if (hf.hl < 16) { if (hf.hl < head) j = hf.hl; }
    else if (hf.hl - 16 < tail) j = S - tail + (hf.hl - 16);
    if (j >= 0) {
      const uint32_t q = mo + (uint32_t)j;
      g[j] = (uint8_t)((bs[q >> 5] >> (q & 31u)) & 1u);
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

// Lanes 0-10 of a half flood the stones of colour c0 from liberty classes 0-10, lanes 11-21 the stones of c1.  With
// `areas`, lanes 22 / 23 additionally flood the EMPTY points from the neighbours of the c0 / c1 stones (Tromp-Taylor
// territory, gym_go/gogame.py:275-300) - same instructions, different inputs, so area scoring rides along for free.
__device__ __forceinline__ Half make_half(int lane, int N, uint32_t inv, bool areas = false) {
  Half hf;
  hf.lane = lane; hf.h = lane >> 5; hf.hl = lane & 31;
  hf.N = N; hf.P = N * N; hf.inv = inv;
  hf.full_l1 = hf.hl < N ? (1u << N) - 1u : 0u;
  const bool second = hf.hl >= kCwClasses;
  hf.cls = hf.hl < kCwLanes ? (hf.hl % kCwClasses) : kCwClasses;
  hf.m_off = second ? 32 : 0;
  hf.r_off = 64 + (second ? 32 : 0);
  hf.s_off = 128;
  if (areas && (hf.hl == kCwLanes || hf.hl == kCwLanes + 1)) {
    hf.cls = kCwClasses + 1;
    hf.m_off = 128; hf.r_off = 160;
    hf.s_off = hf.hl == kCwLanes ? 0 : 32;
  }
  return hf;
}

// The class table, constant memory -> LDS.  A launch's first global reads are a LATENCY chain when each helper waits for its
// own (tools/exp/oneply_where.py: a one-ply launch of k_rollout2<9> spent 2 300 of its 10 900 cycles per wave in a loop of
// five load - wait - ds_write rounds here, another 3 200 in flags -> planes -> generators, one round trip each), so the table
// comes in two halves: cw_table_issue puts ALL its reads in flight, the caller issues whatever else the prologue reads
// (stage_issue_h, flags_issue_h, the generators), and cw_table_commit / stage_commit_h / flags_commit_h consume them in
// issue order - one round trip for the lot.
constexpr int kCwWords = (kCwClasses + 2) * 20, kCwRounds = (kCwWords + kWave - 1) / kWave;
struct CwRegs { uint32_t t[kCwRounds]; };
__device__ __forceinline__ void cw_table_issue(CwRegs &c, int lane) {
  const uint32_t *src = &kCw.m[0][0];
#pragma unroll
  for (int k = 0; k < kCwRounds; ++k) {
    const int i = lane + kWave * k;
    c.t[k] = src[i < (kCwClasses + 1) * 20 ? i : (kCwClasses + 1) * 20 - 1];
  }
}
template <int R>
__device__ __forceinline__ void cw_table_commit(uint32_t *lds, const CwRegs &c, int lane) {
  uint32_t *cwt = lds + Lds2<R>::kCwt;
#pragma unroll
  for (int k = 0; k < kCwRounds; ++k) {
    const int i = lane + kWave * k;
    if (i < kCwWords) cwt[i] = i < (kCwClasses + 1) * 20 ? c.t[k] : 0xFFFFFFFFu;
  }
  WAVE_SYNC();
}
// (the sixteen-board kernels keep the bare class rows: [kCwClasses + 1][20] words)
__device__ __forceinline__ void cw_rows_commit(uint32_t *cwt, const CwRegs &c, int lane) {
#pragma unroll
  for (int k = 0; k < kCwRounds; ++k) {
    const int i = lane + kWave * k;
    if (i < (kCwClasses + 1) * 20) cwt[i] = c.t[k];
  }
}
template <int R>
__device__ __forceinline__ void load_cw_table(uint32_t *lds, int lane) {
  CwRegs c;
  cw_table_issue(c, lane);
  cw_table_commit<R>(lds, c, lane);
}

// stage_in_h / load_flags_h in two halves (see cw_table_issue): the aligned 16-byte vectors of one half's board slice into
// registers (a lane past the slice re-reads its last vector: always mapped, never stored), then registers -> LDS
template <int R>
struct StageRegs {
  static constexpr int NV = (((15 + 4 * R * R + 15) >> 4) + 31) / 32;
  typedef uint32_t Vec __attribute__((ext_vector_type(4)));   // (a register quadruple: a struct would live in scratch)
  Vec v[NV];
};
template <int R>
__device__ __forceinline__ void stage_issue_h(const uint8_t *g, int nbytes, int hl, StageRegs<R> &s) {
  const uint32_t mis = (uint32_t)((uintptr_t)g & 15u);
  const uint8_t *ga = g - mis;
  const int nv = (int)(mis + nbytes + 15) >> 4;
#pragma unroll
  for (int k = 0; k < StageRegs<R>::NV; ++k) {
    const int v = hl + 32 * k;
    s.v[k] = *reinterpret_cast<const typename StageRegs<R>::Vec *>(ga + 16 * (v < nv ? v : nv - 1));
  }
}
template <int R>
__device__ __forceinline__ uint32_t stage_commit_h(const uint8_t *g, int nbytes, uint8_t *lds, int hl, const StageRegs<R> &s) {
  const uint32_t mis = (uint32_t)((uintptr_t)g & 15u);
  const int nv = (int)(mis + nbytes + 15) >> 4;
#pragma unroll
  for (int k = 0; k < StageRegs<R>::NV; ++k) {
    const int v = hl + 32 * k;
    if (v < nv) *reinterpret_cast<typename StageRegs<R>::Vec *>(lds + 16 * v) = s.v[k];
  }
  return mis;
}
__device__ __forceinline__ uint8_t flags_issue_h(const uint8_t *g, int P, int pt, const Half &hf) {
  const int l = hf.hl & 3;
  return g[l == 0 ? 2 * P : l == 1 ? 3 * P + pt : l == 2 ? 4 * P : 5 * P];   // (every lane reads one of the four bytes: no branch)
}
__device__ __forceinline__ uint32_t flags_commit_h(uint8_t fb, const Half &hf) {
  return half_of(__ballot(fb != 0 && hf.hl < 4), hf.h) & 0xFu;
}

// One pair's byte planes: pair_issue puts the half's flag bytes and plane vectors in flight - and, on a wave's first pair
// (`tables` false), the class table; the caller adds its own reads (actions, generators, ko points); pair_commit builds the
// tables if they are still missing (the spread table `lut` when the kernel emits boards), hands the flags back and leaves the
// planes at io + (return value).
template <int R>
struct PairRegs { StageRegs<R> sr; CwRegs cw; uint8_t fb; };
template <int R>
__device__ __forceinline__ void pair_issue(PairRegs<R> &pr, const uint8_t *g, int nbytes, const Half &hf, bool tables) {
  if (!tables) cw_table_issue(pr.cw, hf.lane);
  pr.fb = flags_issue_h(g, hf.P, 0, hf);
  stage_issue_h<R>(g, nbytes, hf.hl, pr.sr);
}
template <int R>
__device__ __forceinline__ uint32_t pair_commit(const PairRegs<R> &pr, const uint8_t *g, int nbytes, uint8_t *io, const Half &hf,
                                                uint32_t *lds, uint2 *lut, bool &tables, uint32_t &flags) {
  if (!tables) {
    if (lut) load_spread_lut(lut, hf.lane);
    cw_table_commit<R>(lds, pr.cw, hf.lane);
    tables = true;
  }
  flags = flags_commit_h(pr.fb, hf);
  WAVE_SYNC();
  const uint32_t mi = stage_commit_h<R>(g, nbytes, io, hf.hl, pr.sr);
  WAVE_SYNC();
  return mi;
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

// ---------------------------------------------------------------- packed boards (3 N + 1 words, see k_pack) as kernel I/O
// The L1 rows ARE the packed format, so the PACKED instantiations of the step kernels skip the byte-plane
// conversions altogether: a board is 232 B instead of 2 166 B and costs no VALU work to read or write.
// fw: bit 0 turn, bit 1 previous move was a pass, bit 2 game over.
__device__ __forceinline__ void load_packed_h(const uint32_t *gp, int N, const Half &hf, uint32_t &black, uint32_t &white,
                                              uint32_t &invalid, uint32_t &fw) {
  black = white = invalid = 0;
  if (hf.hl < N) {
    black = gp[hf.hl];
    white = gp[N + hf.hl];
    invalid = gp[2 * N + hf.hl];
  }
  fw = gp[3 * N];
}
__device__ __forceinline__ void store_packed_h(uint32_t *gp, int N, const Half &hf, uint32_t black, uint32_t white,
                                               uint32_t invalid, uint32_t turn, uint32_t passed, uint32_t done, bool wr) {
  if (wr && hf.hl < N) {
    gp[hf.hl] = black;
    gp[N + hf.hl] = white;
    gp[2 * N + hf.hl] = invalid;
  }
  if (wr && hf.hl == 31) gp[3 * N] = turn | (passed << 1) | (done << 2);
}

// gogame.batch_next_states on packed boards (out of place), two boards per wave
template <int R, bool FULLN = false>
__global__ __launch_bounds__(kWave, 4) void k_next_states_p(const uint32_t *__restrict__ in, const int32_t *__restrict__ actions,
                                                            uint32_t *__restrict__ out, int32_t *__restrict__ status,
                                                            int64_t B, int N, uint32_t inv, int canonical, AgeSplit age) {
  if (FULLN) { N = R; inv = (65536u + R - 1u) / R; }   // N == R: compile-time constants (see k_env_step2)
  __shared__ __attribute__((aligned(16))) uint32_t lds[Lds2<R>::kTotal];
  const Half hf = make_half(threadIdx.x, N, inv);
  load_cw_table<R>(lds, hf.lane);
  const int W = 3 * N + 1;
  const int64_t npairs = (B + 1) >> 1;
  const PairSpan span = pair_span(npairs, age);
  for (int64_t p = span.first; p < span.end; p += span.stride) {
    const int64_t b0 = 2 * p + hf.h;
    const bool on = b0 < B;
    const int64_t b = on ? b0 : B - 1;
    uint32_t black, white, invalid, fw;
    load_packed_h(in + b * (int64_t)W, N, hf, black, white, invalid, fw);
    const int a = actions[b];
    const bool in_range = a >= 0 && a <= hf.P;
    const bool is_pass = a == hf.P;
    bool illegal = !in_range;
    if (in_range && !is_pass) {
      int ar, ac;
      split_action(a, N, hf.inv, ar, ac);
      illegal = (((uint32_t)__shfl((int)invalid, (hf.lane & 32) + ar) >> ac) & 1u) != 0;
    }
    const int pl = fw & 1u;
    uint32_t nb = black, nw = white, ninv = invalid, nturn = pl, passed = (fw >> 1) & 1u, done = (fw >> 2) & 1u;
    if (__ballot(!illegal)) {
      uint32_t mine = pl ? white : black, opp = pl ? black : white;
      uint32_t atari_unused;
      // an illegal half still runs the (wave-wide) analysis on a harmless pass, its result is discarded
      const uint32_t iv = step_core2<R, false>(mine, opp, illegal ? hf.P : a, hf, lds, 0u, false, atari_unused);
      if (!illegal) {
        ninv = iv;
        nb = pl ? opp : mine;
        nw = pl ? mine : opp;
        done = (done || (is_pass && passed)) ? 1u : 0u;
        passed = is_pass ? 1u : 0u;
        nturn = 1u - (uint32_t)pl;
        if (canonical && nturn == 1u) {
          const uint32_t t = nb; nb = nw; nw = t;
          nturn = 0u;
        }
      }
    }
    store_packed_h(out + b * (int64_t)W, N, hf, nb, nw, ninv, nturn, passed, done, on);  // illegal: row passes through
    if (status && on && hf.hl == 0) status[b] = illegal ? GG_STATUS_ILLEGAL : GG_STATUS_OK;
  }
}

// ---------------------------------------------------------------- software-pipelined board I/O (LDS-DMA)
// A per-ply kernel is a chain  load board -> analyse -> store board  per wave, and with 3-4 waves per SIMD the HBM
// round trip of the load (and the acknowledgement of the store) is NOT hidden by the other waves: bytes in flight =
// waves x one board, far below bandwidth x latency.  The loads therefore go global -> LDS directly
// (global_load_lds_dwordx4: no VGPRs are held while they fly, so the next pair's board can be in flight during the
// whole analysis of this pair) and the stores of the previous pair are issued right after them, before the analysis.
// One `s_waitcnt vmcnt(0)` at the top of each iteration then finds everything issued a full analysis ago.
// The DMA is inline asm on purpose: the compiler makes every later ds_read wait for an LDS-DMA it knows about.
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
// lane L's 16 (4) bytes at g land at LDS address m0 + 16 L (m0 + 4 L); m0 must be wave-uniform.  M0 is written and
// consumed inside one asm statement and declared clobbered (clang warns that M0 is a reserved register: silenced here).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16(const void *g, uint32_t m0) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0), "v"(g) : "memory", "m0");
}
__device__ __forceinline__ void dma4(const void *g, uint32_t m0) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(m0), "v"(g) : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// aligned superset of g[0 .. nbytes) -> this half's landing buffer (ROUNDS * 512 bytes per half, half 1 right after
// half 0); board byte j ends up at stage_half + (g & 15) + j, exactly like stage_in_h
template <int ROUNDS>
__device__ __forceinline__ void dma_stage_h(const uint8_t *g, int nbytes, uint32_t stage_lds, const Half &hf) {
  const uint32_t mis = (uint32_t)((uintptr_t)g & 15u);
  const uint8_t *ga = g - mis;
  const int nv = (int)(mis + nbytes + 15) >> 4;
#pragma unroll
  for (int k = 0; k < ROUNDS; ++k) {
    const int v = hf.hl + 32 * k;
    if (hf.h == 0) { if (v < nv) dma16(ga + 16 * v, stage_lds + 512u * k); }
    else { if (v < nv) dma16(ga + 16 * v, stage_lds + 512u * (ROUNDS + k) - 512u); }
  }
}

// gogame.batch_next_states, two boards per wave, pipelined (see above).  Per pair the DMA brings planes 0-1, the four
// flag bytes (as aligned dwords, lanes 0-3 of each half) and the action of the pair after next (lane 4).
template <int R>
__global__ __launch_bounds__(kWave, GG_LB_PLY) void k_next_states2(const uint8_t *__restrict__ in,
                                                        const int32_t *__restrict__ actions,
                                                        uint8_t *__restrict__ out, int32_t *__restrict__ status,
                                                        int64_t B, int N, uint32_t inv, int canonical,
                                                        int cols, uint32_t share1, uint32_t share2) {
  constexpr int kRounds = 2;   // 2 planes + misalignment <= 47 vectors
  __shared__ __attribute__((aligned(16))) uint32_t lds[Lds2<R>::kTotal];
  __shared__ __attribute__((aligned(16))) uint8_t stage[2 * kRounds * 512];
  __shared__ __attribute__((aligned(16))) uint32_t meta[kWave];
  const Half hf = make_half(threadIdx.x, N, inv);
  __shared__ uint2 lut[256];
  CwRegs cw;
  cw_table_issue(cw, hf.lane);   // (in flight with the first action and, behind it, the first pair's DMA: see cw_table_issue)
  const int S = 6 * hf.P;
  const int64_t npairs = (B + 1) >> 1;
  // Which pairs a wave takes.  The grid is the resident set: `cols` SIMDs x 3 waves, and the dispatcher places workgroups
  // c, c + cols, c + 2 cols on the same SIMD in that order (tools/exp/where_ns.py (round 3, in git history): on all 1 024 SIMDs).  The arbiter serves
  // the oldest wave first, so with equal shares the three finish at 0.69 / 0.86 / 1.00 of the launch and the SIMD idles
  // towards the end.  The pairs of a column (c, c + cols, c + 2 cols ...) are therefore split UNEVENLY among its three
  // waves - the oldest takes the fraction share1, the next share2 - share1, the youngest the rest - so that they finish
  // together.  cols == 0 (small batches): one wave per pair stride, as before.
  int64_t p, G, pend;
  if (cols > 0) {
    const int c = (int)(blockIdx.x % (unsigned)cols), r = (int)(blockIdx.x / (unsigned)cols);
    const int64_t K = (npairs - c + cols - 1) / cols;                 // pairs of this column
    const int64_t k1 = (K * share1) >> 16, k2 = (K * share2) >> 16;
    const int64_t a0 = r == 0 ? 0 : r == 1 ? k1 : k2, a1 = r == 0 ? k1 : r == 1 ? k2 : K;
    if (a0 >= a1) return;
    G = cols;
    p = c + G * a0;
    pend = c + G * a1;
    if (pend > npairs) pend = npairs;
  } else {
    G = gridDim.x;
    p = blockIdx.x;
    pend = npairs;
  }
  const uint32_t stage_lds = lds_addr(stage), meta_lds = lds_addr(meta);
  const uint8_t *sh = stage + hf.h * (kRounds * 512);
  uint32_t *work = lds + hf.h * 128;   // emit_store_h scratch (free outside the analysis)

  auto board_of = [&](int64_t p) -> int64_t { const int64_t b0 = 2 * p + hf.h; return b0 < B ? b0 : B - 1; };
  auto flag_off = [&](int a) -> int {   // byte offset inside the board of this lane's flag byte (lanes 0-3)
    const bool pt = a >= 0 && a < hf.P;
    return hf.hl == 0 ? 2 * hf.P : hf.hl == 1 ? 3 * hf.P + (pt ? a : 0) : hf.hl == 2 ? 4 * hf.P : 5 * hf.P;
  };
  auto issue = [&](int64_t p, int a) {   // everything pair p needs + the action of pair p + 2 G
    const uint8_t *gi = in + board_of(p) * (int64_t)S;
    dma_stage_h<kRounds>(gi, 2 * hf.P, stage_lds, hf);
    const int64_t p2 = p + G;
    const uint8_t *fa = gi + flag_off(a);
    const void *src = hf.hl == 4 ? (const void *)(actions + board_of(p2 < pend ? p2 : p))
                                 : (const void *)(fa - ((uintptr_t)fa & 3u));
    if (hf.hl < 5) dma4(src, meta_lds);
  };

  if (p >= pend) return;
#ifdef GG_AB_WHERE
  const long long tw0_ = wall_clock64();
#endif
  int a = actions[board_of(p)];
  issue(p, a);
  load_spread_lut(lut, hf.lane);             // both tables are built while the first pair's DMA is in flight
  cw_table_commit<R>(lds, cw, hf.lane);
  bool have_prev = false;
  uint32_t pb = 0, pw = 0, pi = 0, pm = 0;   // previous pair's result rows + {turn, passed, done, illegal} bits
  for (;;) {
    dma_wait();
    WAVE_SYNC();
    const int64_t b0 = 2 * p + hf.h;
    const bool on = b0 < B;
    const int64_t b = on ? b0 : B - 1;
    const uint8_t *gi = in + b * (int64_t)S;
    const uint32_t mi = (uint32_t)((uintptr_t)gi & 15u);
    uint32_t black = plane_to_row<R>(sh + mi, N, hf.hl);
    uint32_t white = plane_to_row<R>(sh + mi + hf.P, N, hf.hl);
    const uint32_t mw = meta[hf.lane];
    const uint32_t fsh = 8u * (uint32_t)((uintptr_t)(gi + flag_off(a)) & 3u);
    const uint32_t flags = half_of(__ballot(hf.hl < 4 && ((mw >> fsh) & 0xFFu) != 0), hf.h) & 0xFu;
    const int a_next = __shfl((int)mw, (hf.lane & 32) + 4);
    lds_drain();
    WAVE_SYNC();
    const int64_t pn = p + G;
    if (pn < pend) issue(pn, a_next);
    if (have_prev) {   // stores of the previous pair fly during this pair's analysis
      const int64_t q0 = 2 * (p - G) + hf.h;
      const bool qon = q0 < B;
      const int64_t q = qon ? q0 : B - 1;
      const bool ill = (pm >> 3) & 1u;
      if (__ballot(!ill))
        emit_store_h<R>(out + q * (int64_t)S, pb, pw, pi, pm & 1u, (pm >> 1) & 1u, (pm >> 2) & 1u, hf, work, lut, qon && !ill);
      if (ill) copy_row_h(in + q * (int64_t)S, out + q * (int64_t)S, S, hf.hl, qon);  // rare: row passes through
      if (status && qon && hf.hl == 0) status[q] = ill ? GG_STATUS_ILLEGAL : GG_STATUS_OK;
    }
    const bool in_range = a >= 0 && a <= hf.P;
    const bool is_pass = a == hf.P;
    const bool illegal = !in_range || (!is_pass && (flags & 2u));
    const int pl = flags & 1u;
    uint32_t invalid = 0;
    if (__ballot(!illegal)) {
      uint32_t mine = pl ? white : black, opp = pl ? black : white;
      // an illegal half still runs the (wave-wide) analysis on a harmless pass, its result is discarded
      uint32_t atari_unused;
      invalid = step_core2<R, false>(mine, opp, illegal ? hf.P : a, hf, lds, 0u, false, atari_unused);
      black = pl ? opp : mine;
      white = pl ? mine : opp;
    }
    const uint32_t passed = is_pass ? 1 : 0;
    const uint32_t done = ((flags & 8u) || (is_pass && (flags & 4u))) ? 1 : 0;
    int nturn = 1 - pl;
    if (canonical && nturn == 1) {
      uint32_t t = black; black = white; white = t;
      nturn = 0;
    }
    pb = black; pw = white; pi = invalid;
    pm = (uint32_t)nturn | (passed << 1) | (done << 2) | ((illegal ? 1u : 0u) << 3);
    have_prev = true;
    a = a_next;
    if (pn >= pend) break;
    p = pn;
  }
  {
    const int64_t q0 = 2 * p + hf.h;
    const bool qon = q0 < B;
    const int64_t q = qon ? q0 : B - 1;
    const bool ill = (pm >> 3) & 1u;
    WAVE_SYNC();
    if (__ballot(!ill))
      emit_store_h<R>(out + q * (int64_t)S, pb, pw, pi, pm & 1u, (pm >> 1) & 1u, (pm >> 2) & 1u, hf, work, lut, qon && !ill);
    if (ill) copy_row_h(in + q * (int64_t)S, out + q * (int64_t)S, S, hf.hl, qon);
    if (status && qon && hf.hl == 0) status[q] = ill ? GG_STATUS_ILLEGAL : GG_STATUS_OK;
  }
#ifdef GG_AB_WHERE
  if (threadIdx.x == 0 && blockIdx.x < 16384) {
    unsigned int hw_, xc_;
    asm volatile("s_getreg_b32 %0, hwreg(4)" : "=s"(hw_));
    asm volatile("s_getreg_b32 %0, hwreg(20)" : "=s"(xc_));
    gg_where[3 * blockIdx.x] = xc_; gg_where[3 * blockIdx.x + 1] = hw_; gg_where[3 * blockIdx.x + 2] = (unsigned int)(wall_clock64() - tw0_);
  }
#endif
}

// gogame.batch_next_states on a batch that leaves the machine under-filled (<= two waves per SIMD): one pair per wave, no
// pipeline to fill - so every read of the pair (class table, flag bytes, planes 0 - 3, the action) is in flight together
// (cw_table_issue), the legality of the move is read off the staged invalid-move plane instead of a dependent INVD[action]
// load, and the launch goes out as WPB-wave workgroups (rollout2_body).  Same results as k_next_states2: next_state's
// semantics per game, an illegal / out-of-range move passes its row through with status 1.
template <int R, int WPB>
__global__ __launch_bounds__(kWave * WPB, GG_LB_PLY) void k_next_states2s(const uint8_t *__restrict__ in,
                                                        const int32_t *__restrict__ actions,
                                                        uint8_t *__restrict__ out, int32_t *__restrict__ status,
                                                        int64_t B, int N, uint32_t inv, int canonical) {
  __shared__ __attribute__((aligned(16))) uint32_t lds_[WPB][Lds2<R>::kTotal];
  __shared__ uint2 lut_[WPB][256];
  const int wv = WPB > 1 ? (int)(threadIdx.x >> 6) : 0;
  uint32_t *lds = lds_[wv];
  uint2 *lut = lut_[wv];
  const Half hf = make_half((int)(threadIdx.x & (kWave - 1)), N, inv);
  const int S = 6 * hf.P;
  uint8_t *io = reinterpret_cast<uint8_t *>(lds) + hf.h * Cfg<R>::kIoBytes;
  bool tables = false;
  const int64_t npairs = (B + 1) >> 1;
  for (int64_t p = (int64_t)blockIdx.x * WPB + wv; p < npairs; p += (int64_t)gridDim.x * WPB) {
    const bool on = 2 * p + hf.h < B;
    const int64_t b = on ? 2 * p + hf.h : B - 1;
    const uint8_t *gi = in + b * (int64_t)S;
    uint8_t *go = out + b * (int64_t)S;
    PairRegs<R> pr;
    pair_issue<R>(pr, gi, 4 * hf.P, hf, tables);
    const int a = actions[b];
    uint32_t flags;   // bit 0 turn, (bit 1: not used here), bit 2 previous move was a pass, bit 3 game over
    const uint32_t mi = pair_commit<R>(pr, gi, 4 * hf.P, io, hf, lds, lut, tables, flags);
    uint32_t black = plane_to_row<R>(io + mi, N, hf.hl);
    uint32_t white = plane_to_row<R>(io + mi + hf.P, N, hf.hl);
    const uint32_t invd = plane_to_row<R>(io + mi + 3 * hf.P, N, hf.hl);
    const bool in_range = a >= 0 && a <= hf.P;
    const bool is_pass = a == hf.P;
    int ar = 0, ac = 0;
    if (in_range && !is_pass) split_action(a, N, hf.inv, ar, ac);
    // the lane that holds row ar of the half's board tests the mask bit (gym_go/gogame.py:59), the half shares the verdict
    const bool bad = half_of(__ballot(in_range && !is_pass && hf.hl == ar && ((invd >> ac) & 1u)), hf.h) != 0;
    const bool illegal = !in_range || bad;
    const int pl = flags & 1u;
    uint32_t invalid = 0;
    if (__ballot(!illegal)) {
      uint32_t mine = pl ? white : black, opp = pl ? black : white;
      uint32_t atari_unused;   // (an illegal half runs the wave-wide analysis on a harmless pass, its result is discarded)
      invalid = step_core2<R, false>(mine, opp, illegal ? hf.P : a, hf, lds, 0u, false, atari_unused);
      black = pl ? opp : mine;
      white = pl ? mine : opp;
    }
    const uint32_t passed = is_pass ? 1 : 0;
    const uint32_t done = ((flags & 8u) || (is_pass && (flags & 4u))) ? 1 : 0;
    int nturn = 1 - pl;
    if (canonical && nturn == 1) {
      const uint32_t t = black; black = white; white = t;
      nturn = 0;
    }
    WAVE_SYNC();
    if (__ballot(!illegal))
      emit_store_h<R>(go, black, white, invalid, (uint32_t)nturn, passed, done, hf, reinterpret_cast<uint32_t *>(io), lut, on && !illegal);
    if (illegal) copy_row_h(gi, go, S, hf.hl, on);
    if (status && on && hf.hl == 0) status[b] = illegal ? GG_STATUS_ILLEGAL : GG_STATUS_OK;
  }
}

// PERPLY = instantiation for 1-2 plies per launch: the board I/O dominates there and the kernel runs best spill-free
// at 3 waves per SIMD; the fused instantiation keeps its hot ply loop spill-free at 4 waves per SIMD.
// PACKED: `states` holds packed boards (uint32 [B][3 N + 1]).
// WPB: waves per workgroup (each wave an independent pair of boards with its own LDS).  A launch of a few thousand single-wave
// workgroups is DISPATCHED over ~0.26 ns per workgroup (tools/exp/oneply_ramp.py: 2 048 workgroups enter the machine over
// 0.54 us, 512 over 0.22), which is a tenth of a one-ply launch of config 2's size: small launches go out as WPB-wave workgroups.
template <int R, bool PERPLY, bool PACKED, bool FULLN, int WPB>
__device__ __forceinline__ void rollout2_body(uint8_t *__restrict__ states, uint64_t *__restrict__ rng,
                                              int32_t *__restrict__ last_actions,
                                              int64_t *__restrict__ steps_done, int64_t B, int N, uint32_t inv,
                                              int plies, int auto_reset, const AgeSplit &age) {
  if (FULLN) { N = R; inv = (65536u + R - 1u) / R; }   // N == R: compile-time constants (see k_env_step2)
  __shared__ __attribute__((aligned(16))) uint32_t lds_[WPB][Lds2<R>::kTotal];
  __shared__ uint32_t fair_mates_[WPB][16];
  __shared__ uint2 lut_[WPB][256];
  const int wv = WPB > 1 ? (int)(threadIdx.x >> 6) : 0;
  uint32_t *lds = lds_[wv], *fair_mates = fair_mates_[wv];
  uint2 *lut = lut_[wv];
  GG_PROF_DECL;
  const Half hf = make_half((int)(threadIdx.x & (kWave - 1)), N, inv);
  // the prologue's global reads - class table, the first pair's flags, planes and generators - are in flight TOGETHER
  // (cw_table_issue): one round trip instead of eight (9x9 x 4 096 games, one ply per launch as a hipGraph node: 7.15 -> 6.07 us)
  bool tables = false;
  const int S = 6 * hf.P;
  uint8_t *io = reinterpret_cast<uint8_t *>(lds) + hf.h * Cfg<R>::kIoBytes;
  const int64_t npairs = (B + 1) >> 1;
  PairSpan span = pair_span(npairs, age);
  if (WPB > 1) {   // (launched without an age split: wave w of workgroup g is "workgroup" g WPB + w of the single-wave form)
    span.first = (int64_t)blockIdx.x * WPB + wv;
    span.stride = (int64_t)gridDim.x * WPB;
    span.end = npairs;
  }
  for (int64_t p = span.first; p < span.end; p += span.stride) {
    const int64_t bA = 2 * p, bB = (2 * p + 1 < B) ? 2 * p + 1 : B - 1;
    const bool on = 2 * p + hf.h < B;
    const int64_t b = hf.h ? bB : bA;
    uint8_t *gs = states + b * (int64_t)S;
    uint32_t *gp = reinterpret_cast<uint32_t *>(states) + b * (int64_t)(3 * N + 1);
    uint32_t black, white, invalid;
    int turn, passed, done;
    PairRegs<R> pr;
    if (PACKED) {
      if (!tables) { load_cw_table<R>(lds, hf.lane); load_spread_lut(lut, hf.lane); tables = true; }
    } else {
      pair_issue<R>(pr, gs, 4 * hf.P, hf, tables);
    }
    const uint64_t ra = rng[bA], rb = rng[bB];
    if (PACKED) {
      uint32_t fw;
      load_packed_h(gp, N, hf, black, white, invalid, fw);
      turn = fw & 1u; passed = (fw >> 1) & 1u; done = (fw >> 2) & 1u;
    } else {
      uint32_t flags;
      const uint32_t mi = pair_commit<R>(pr, gs, 4 * hf.P, io, hf, lds, lut, tables, flags);
      black = plane_to_row<R>(io + mi, N, hf.hl);
      white = plane_to_row<R>(io + mi + hf.P, N, hf.hl);
      invalid = plane_to_row<R>(io + mi + 3 * hf.P, N, hf.hl);
      turn = flags & 1u; passed = (flags >> 2) & 1u; done = (flags >> 3) & 1u;
    }
    GG_PROF(5);   // tables + load
    uint64_t xa = uniform64(ra), xb = uniform64(rb);  // generator states live in SGPRs
    int last = -1, played = 0;
    uint32_t atari = 0;   // next mover's opponents in atari, known from the previous ply of this launch
    bool have_atari = false;
    FairShare fair(fair_mates, !PERPLY);   // a fused launch: the waves of a SIMD advance together (gg_common.h)
#pragma unroll 1
    for (int t = 0; t < plies; ++t) {
      if (!PERPLY && (t & 3) == 0 && plies >= 8) {   // (the band of k_rollout4: never wider than the plies that are left)
        const uint32_t band = plies >= 192 ? 24u : (plies >= 16 ? (uint32_t)plies >> 3 : 2u), left = (uint32_t)(plies - t);
        fair.update((uint32_t)t, left < band ? (left > 2u ? left : 2u) : band);
      }
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
      GG_PROF(0);
      uint32_t ninv = step_core2<R, false>(mine, opp, live ? a : hf.P, hf, lds, atari, have_atari, natari);
      GG_PROF(4);
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
    if (!PERPLY && plies >= 8) fair.release();
    if (PACKED) {
      store_packed_h(gp, N, hf, black, white, invalid, (uint32_t)turn, (uint32_t)passed, (uint32_t)done, on && played != 0);
    } else if (__ballot(played != 0)) {
      emit_store_h<R>(gs, black, white, invalid, (uint32_t)turn, (uint32_t)passed, (uint32_t)done, hf,
                      reinterpret_cast<uint32_t *>(io), lut, on && played != 0);
    }
    if (on && hf.hl == 0) {
      rng[b] = hf.h ? xb : xa;
      if (last_actions) last_actions[b] = last;
      if (steps_done && played) atomicAdd(reinterpret_cast<unsigned long long *>(steps_done) + b, (unsigned long long)played);
    }
    GG_PROF(7);   // write-back
  }
  GG_PROF_FLUSH;
}
template <int R, bool PERPLY, bool PACKED = false, bool FULLN = false>
__global__ __launch_bounds__(kWave, (PERPLY && !PACKED) ? GG_LB_PLY : 4) void k_rollout2(uint8_t *__restrict__ states, uint64_t *__restrict__ rng,
                                                    int32_t *__restrict__ last_actions,
                                                    int64_t *__restrict__ steps_done, int64_t B, int N, uint32_t inv,
                                                    int plies, int auto_reset, AgeSplit age) {
  rollout2_body<R, PERPLY, PACKED, FULLN, 1>(states, rng, last_actions, steps_done, B, N, inv, plies, auto_reset, age);
}
// one / two plies per launch on byte planes, four waves per workgroup (WPB above)
template <int R, bool FULLN>
__global__ __launch_bounds__(4 * kWave, GG_LB_PLY) void k_rollout2_w4(uint8_t *__restrict__ states, uint64_t *__restrict__ rng,
                                                    int32_t *__restrict__ last_actions,
                                                    int64_t *__restrict__ steps_done, int64_t B, int N, uint32_t inv,
                                                    int plies, int auto_reset, AgeSplit age) {
  rollout2_body<R, true, false, FULLN, 4>(states, rng, last_actions, steps_done, B, N, inv, plies, auto_reset, age);
}

// Replay of given move sequences with the boards resident on-chip: state = next_state(state, moves[b][t]) for
// t = 0 .. T-1 (a loop of gogame.next_state / GoEnv.step: gym_go/gogame.py:34-87, gym_go/envs/go_env.py:49-76) in ONE
// launch.  A game stops at its first move that is out of range, on an invalid point (gogame.py:59) or made after the game
// has ended (go_env.py:53); played[b] = number of moves applied = index of that move, or T.  moves: int32 [B][T].
template <int R, bool PACKED>
__global__ __launch_bounds__(kWave, 4) void k_play_moves2(uint8_t *__restrict__ states, const int32_t *__restrict__ moves,
                                                          int32_t *__restrict__ played_out, int64_t B, int N, uint32_t inv,
                                                          int T, AgeSplit age) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[Lds2<R>::kTotal];
  const Half hf = make_half(threadIdx.x, N, inv);
  __shared__ uint2 lut[256];
  bool tables = false;
  const int S = 6 * hf.P;
  uint8_t *io = reinterpret_cast<uint8_t *>(lds) + hf.h * Cfg<R>::kIoBytes;
  const int64_t npairs = (B + 1) >> 1;
  const PairSpan span = pair_span(npairs, age);
  for (int64_t p = span.first; p < span.end; p += span.stride) {
    const bool on = 2 * p + hf.h < B;
    const int64_t b = on ? 2 * p + hf.h : B - 1;
    uint8_t *gs = states + b * (int64_t)S;
    uint32_t *gp = reinterpret_cast<uint32_t *>(states) + b * (int64_t)(3 * N + 1);
    uint32_t black, white, invalid;
    int turn, passed, done;
    if (PACKED) {
      if (!tables) { load_cw_table<R>(lds, hf.lane); load_spread_lut(lut, hf.lane); tables = true; }
      uint32_t fw;
      load_packed_h(gp, N, hf, black, white, invalid, fw);
      turn = fw & 1u; passed = (fw >> 1) & 1u; done = (fw >> 2) & 1u;
    } else {
      PairRegs<R> pr;
      pair_issue<R>(pr, gs, 4 * hf.P, hf, tables);
      uint32_t flags;
      const uint32_t mi = pair_commit<R>(pr, gs, 4 * hf.P, io, hf, lds, lut, tables, flags);
      black = plane_to_row<R>(io + mi, N, hf.hl);
      white = plane_to_row<R>(io + mi + hf.P, N, hf.hl);
      invalid = plane_to_row<R>(io + mi + 3 * hf.P, N, hf.hl);
      turn = flags & 1u; passed = (flags >> 2) & 1u; done = (flags >> 3) & 1u;
    }
    int played = 0;
    bool live = on;
    uint32_t atari = 0;
    bool have_atari = false;
    const int32_t *mv_row = moves + b * (int64_t)T;
#pragma unroll 1
    for (int t0 = 0; t0 < T && __ballot(live); t0 += 32) {
      const int mv = (t0 + hf.hl < T) ? mv_row[t0 + hf.hl] : -1;   // lane l of a half holds move t0 + l of its game
      const int nt = T - t0 < 32 ? T - t0 : 32;
#pragma unroll 1
      for (int i = 0; i < nt; ++i) {
        if (__ballot(live) == 0) break;
        const int a = __shfl(mv, (hf.lane & 32) + i);
        bool legal = live && !done && a >= 0 && a <= hf.P;
        if (legal && a < hf.P) {
          int ar, ac;
          split_action(a, N, hf.inv, ar, ac);
          // (the condition is uniform inside a half and the source lane is in the same half, hence active)
          legal = (((uint32_t)__shfl((int)invalid, (hf.lane & 32) + ar) >> ac) & 1u) == 0;
        }
        live = legal;
        uint32_t mine = turn ? white : black, opp = turn ? black : white;
        uint32_t natari;
        // a stopped half still runs the (wave-wide) analysis on a harmless pass, its result is discarded
        const uint32_t ninv = step_core2<R, false>(mine, opp, legal ? a : hf.P, hf, lds, atari, have_atari, natari);
        have_atari = true;
        if (legal) {
          atari = natari;
          invalid = ninv;
          black = turn ? opp : mine;
          white = turn ? mine : opp;
          if (a == hf.P) { if (passed) done = 1; passed = 1; } else passed = 0;
          turn ^= 1;
          ++played;
        }
      }
    }
    if (PACKED) {
      store_packed_h(gp, N, hf, black, white, invalid, (uint32_t)turn, (uint32_t)passed, (uint32_t)done, on && played != 0);
    } else if (__ballot(played != 0)) {
      emit_store_h<R>(gs, black, white, invalid, (uint32_t)turn, (uint32_t)passed, (uint32_t)done, hf,
                      reinterpret_cast<uint32_t *>(io), lut, on && played != 0);
    }
    if (played_out && on && hf.hl == 0) played_out[b] = played;
  }
}

// One GoEnv.step for every game of a batched env, in place, one launch (gym_go/envs/go_env.py:49-76):
// auto-reset of finished games (:40-47), the action (given, or drawn like uniform_random_action :78-81), the legality
// check (gogame.py:59), next_state, game_ended and GoEnv.reward (:128-149; Tromp-Taylor areas gogame.py:275-300).
// HEUR (reward_method heuristic: the area margin is the reward of every ply): the two area floods of the post-move
// position ride in the idle flood lanes of the liberty analysis.  !HEUR (reward_method real): the areas only matter
// when a game ends, so the step runs the plain analysis and a wave whose pair just finished a game (two passes: no
// stone moved, the liberty classes are not needed again) runs one more analysis for the territory.
template <int R, bool HEUR, bool PACKED, bool FULLN, int WPB>
__device__ __forceinline__ void env_step2_body(uint8_t *__restrict__ states, const int32_t *__restrict__ actions,
                                               uint64_t *__restrict__ rng, float *__restrict__ rewards,
                                               uint8_t *__restrict__ dones, int32_t *__restrict__ status,
                                               int32_t *__restrict__ taken, int64_t B, int N, uint32_t inv,
                                               float komi, int auto_reset, const AgeSplit &age,
                                               int32_t *__restrict__ areas = nullptr, int real_formula = 0) {
  // areas (HEUR instantiations, nullable): int32 [B][2], the black / white Tromp-Taylor areas of the resulting position
  // (gg_batch_env_step_scored: GoEnv.step in ONE launch); real_formula: the HEUR instantiation (areas every step) pays
  // out GG_REWARD_REAL's reward
  // FULLN: the board fills the row capacity (N == R) - N, N * N and the reciprocal become compile-time constants
  // (GoVecEnv.step 19x19: 7.9e8 against 7.6e8 steps/s; the same on k_next_states2 costs registers: 7.0e8 against 1.0e9)
  if (FULLN) { N = R; inv = (65536u + R - 1u) / R; }
  __shared__ __attribute__((aligned(16))) uint32_t lds_[WPB][Lds2<R>::kTotal];
  __shared__ uint2 lut_[WPB][256];
  const int wv = WPB > 1 ? (int)(threadIdx.x >> 6) : 0;   // (WPB waves per workgroup: rollout2_body)
  uint32_t *lds = lds_[wv];
  uint2 *lut = lut_[wv];
  const Half hf = make_half((int)(threadIdx.x & (kWave - 1)), N, inv, HEUR);
  const Half hfa = make_half((int)(threadIdx.x & (kWave - 1)), N, inv, true);   // lanes 22 / 23 of each half flood the empty points
  bool tables = false;
  const int S = 6 * hf.P;
  uint8_t *io = reinterpret_cast<uint8_t *>(lds) + hf.h * Cfg<R>::kIoBytes;
  const int64_t npairs = (B + 1) >> 1;
  PairSpan span = pair_span(npairs, age);
  if (WPB > 1) {
    span.first = (int64_t)blockIdx.x * WPB + wv;
    span.stride = (int64_t)gridDim.x * WPB;
    span.end = npairs;
  }
  for (int64_t p = span.first; p < span.end; p += span.stride) {
    const int64_t bA = 2 * p, bB = (2 * p + 1 < B) ? 2 * p + 1 : B - 1;
    const bool on = 2 * p + hf.h < B;
    const int64_t b = hf.h ? bB : bA;
    uint8_t *gs = states + b * (int64_t)S;
    uint32_t *gp = reinterpret_cast<uint32_t *>(states) + b * (int64_t)(3 * N + 1);
    uint32_t black, white, invalid;
    int turn, passed, done;
    PairRegs<R> pr;
    if (PACKED) {
      if (!tables) { load_cw_table<R>(lds, hf.lane); load_spread_lut(lut, hf.lane); tables = true; }
    } else {
      pair_issue<R>(pr, gs, 4 * hf.P, hf, tables);
    }
    // (the given action or the generators: in flight with the planes)
    const int a_given = actions ? actions[b] : 0;
    const uint64_t rga = actions ? 0 : rng[bA], rgb = actions ? 0 : rng[bB];
    if (PACKED) {
      uint32_t fw;
      load_packed_h(gp, N, hf, black, white, invalid, fw);
      turn = fw & 1u; passed = (fw >> 1) & 1u; done = (fw >> 2) & 1u;
    } else {
      uint32_t flags;
      const uint32_t mi = pair_commit<R>(pr, gs, 4 * hf.P, io, hf, lds, lut, tables, flags);
      black = plane_to_row<R>(io + mi, N, hf.hl);
      white = plane_to_row<R>(io + mi + hf.P, N, hf.hl);
      invalid = plane_to_row<R>(io + mi + 3 * hf.P, N, hf.hl);
      turn = flags & 1u; passed = (flags >> 2) & 1u; done = (flags >> 3) & 1u;
    }
    bool wr = false;               // the stored state changes
    const bool frozen = done && !auto_reset;   // go_env.py:53 "assert not self.done"
    if (done && auto_reset) {
      black = white = invalid = 0;
      turn = passed = done = 0;
      wr = true;
    }
    int a;
    if (actions) {
      a = a_given;
    } else {
      uint64_t xa = uniform64(rga), xb = uniform64(rgb);
      const uint32_t valid = hf.full_l1 & ~invalid;
      const uint32_t incl = half_scan((uint32_t)__popc(valid));
      const uint32_t cnt_a = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31);
      const uint32_t cnt_b = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      const uint64_t ua = splitmix_next(xa), ub = splitmix_next(xb);
      const uint32_t ka = (uint32_t)(((ua >> 32) * (uint64_t)(cnt_a + 1)) >> 32);
      const uint32_t kb = (uint32_t)(((ub >> 32) * (uint64_t)(cnt_b + 1)) >> 32);
      a = pick_action2(valid, incl, hf.h ? kb : ka, hf);
      if (on && !frozen && hf.hl == 0) rng[b] = hf.h ? xb : xa;
    }
    const bool in_range = a >= 0 && a <= hf.P;
    bool bad = !in_range || frozen;
    if (in_range && a < hf.P) {
      int ar, ac;
      split_action(a, N, hf.inv, ar, ac);
      const uint32_t row = (uint32_t)__shfl((int)invalid, (hf.lane & 32) + ar);
      bad = bad || ((row >> ac) & 1u);
    }
    uint32_t mine = turn ? white : black, opp = turn ? black : white;
    uint32_t atari_unused, reach[2] = {0u, 0u};
    // a refused half still runs the (wave-wide) analysis on a harmless pass: it yields the areas of its position
    const uint32_t ninv = step_core2<R, false, HEUR>(mine, opp, bad ? hf.P : a, hf, lds, 0u, false, atari_unused, reach);
    if (!bad) {
      invalid = ninv;
      black = turn ? opp : mine;
      white = turn ? mine : opp;
      if (a == hf.P) { if (passed) done = 1; passed = 1; } else passed = 0;
      turn ^= 1;
      wr = true;
    }
    if (!HEUR && __ballot(on && done)) {   // a game of the pair is over (or was refused as over): score it
      uint32_t m0, a0, m1;
      analyze2<R, false, true>(opp, mine, hf.full_l1 & ~(mine | opp), hfa, lds, m0, a0, m1, reach);
    }
    // analyze2 saw (c0, c1) = (opp, mine) of the mover: reach[0] = empties touching opp's colour, reach[1] = mine's
    const bool mover_white = bad ? turn : !turn;   // turn was flipped on commit
    const uint32_t rb = mover_white ? reach[0] : reach[1], rw = mover_white ? reach[1] : reach[0];
    const uint32_t ab = half_scan((uint32_t)(__popc(black) + __popc(rb & ~rw)));
    const uint32_t aw = half_scan((uint32_t)(__popc(white) + __popc(rw & ~rb)));
    if (PACKED) {
      store_packed_h(gp, N, hf, black, white, invalid, (uint32_t)turn, (uint32_t)passed, (uint32_t)done, on && wr);
    } else if (__ballot(wr)) {
      emit_store_h<R>(gs, black, white, invalid, (uint32_t)turn, (uint32_t)passed, (uint32_t)done, hf,
                      reinterpret_cast<uint32_t *>(io), lut, on && wr);
    }
    if (on && hf.hl == 31) {
      const float margin = (float)((int)ab - (int)aw) - komi;
      float rwd;
      if (HEUR && !real_formula) rwd = done ? (margin > 0.f ? 1.f : -1.f) * (float)hf.P : margin;
      else rwd = done ? (margin > 0.f ? 1.f : margin < 0.f ? -1.f : 0.f) : 0.f;
      if (rewards) rewards[b] = rwd;
      if (dones) dones[b] = (uint8_t)done;
      if (status) status[b] = bad ? GG_STATUS_ILLEGAL : GG_STATUS_OK;
      if (taken) taken[b] = a;
      if (HEUR && areas) { areas[2 * b] = (int32_t)ab; areas[2 * b + 1] = (int32_t)aw; }
    }
  }
}
template <int R, bool HEUR, bool PACKED = false, bool FULLN = false>
__global__ __launch_bounds__(kWave, PACKED ? 4 : GG_LB_PLY) void k_env_step2(uint8_t *__restrict__ states, const int32_t *__restrict__ actions,
                                                        uint64_t *__restrict__ rng, float *__restrict__ rewards,
                                                        uint8_t *__restrict__ dones, int32_t *__restrict__ status,
                                                        int32_t *__restrict__ taken, int64_t B, int N, uint32_t inv,
                                                        float komi, int auto_reset, AgeSplit age,
                                                        int32_t *__restrict__ areas = nullptr, int real_formula = 0) {
  env_step2_body<R, HEUR, PACKED, FULLN, 1>(states, actions, rng, rewards, dones, status, taken, B, N, inv, komi, auto_reset, age, areas, real_formula);
}
// byte planes, small batches: four waves per workgroup (rollout2_body)
template <int R, bool HEUR, bool FULLN>
__global__ __launch_bounds__(4 * kWave, GG_LB_PLY) void k_env_step2_w4(uint8_t *__restrict__ states, const int32_t *__restrict__ actions,
                                                        uint64_t *__restrict__ rng, float *__restrict__ rewards,
                                                        uint8_t *__restrict__ dones, int32_t *__restrict__ status,
                                                        int32_t *__restrict__ taken, int64_t B, int N, uint32_t inv,
                                                        float komi, int auto_reset, AgeSplit age,
                                                        int32_t *__restrict__ areas = nullptr, int real_formula = 0) {
  env_step2_body<R, HEUR, false, FULLN, 4>(states, actions, rng, rewards, dones, status, taken, B, N, inv, komi, auto_reset, age, areas, real_formula);
}


// gogame.children (gym_go/gogame.py:175-186), INCREMENTAL: one parent per wave, analysed once, every child derived from it.
//
// Re-analysing every child from scratch costs ~950 VALU per pair of children, which made the kernel VALU-bound at half
// of the HBM-write roofline (round 1).  But a child differs from its parent by one stone plus its captures, and only
// the groups adjacent to the new stone q (and, rarely, to a captured group) change their liberty count:
//   * opponent groups adjacent to q lose exactly the liberty q: count 1 -> captured, 2 -> atari, >= 3 -> still >= 2;
//   * the mover's groups adjacent to q merge with the new stone into G, whose liberties are dilate(G) & empty';
//   * a mover's group in atari next to a captured group gains a liberty (rare; finished by a short L1 flood).
// What this needs from the parent is (a) for every stone the EXACT liberty count of its group, saturated at 3, and
// (b) for every empty point q the stones of each colour whose group has q as a liberty.  Both come out of the same
// floods: lane (h, l) floods colour h (0 = mover, 1 = opponent) from the stones adjacent to the l-th empty point of
// the batch - 32 empty points x 2 colours per wave flood.  Pass 1 runs them for all empty points and counts, per
// stone, how many floods reach it (three OR-AND ops per flood in L1: ge1, ge2, ge3).  Pass 2 runs them again for the
// points of this work item's chunk (the 5 KB transpose buffer holds one batch; when the board has <= 32 empty points
// pass 1's batch is reused) and derives two children per L1 pass, one per half, with ~70 VALU ops + the emitter.
// PACKED: parents and children are packed boards (uint32 [3 N + 1] each): 84 KB instead of 786 KB per 19x19 parent.
// COMPACT (byte planes): gogame.children(..., padded=False) - gym_go/gogame.py:179, the un-padded result the reference
// computes first - for a whole batch: only the slots of the actions valid_moves() keeps (plane 3 clear, + the pass; every
// action once the game has ended, :155-156), in ascending action order, parent b's first child at offsets[b] (exclusive
// scan of the per-parent counts, k_children_counts + k_children_order_scan below).  The same analysis and the same streaming
// emitter: a child's place in the stream is the RANK of its action among the kept ones instead of the action itself, so the
// zero slots of the illegal actions - two thirds of the padded bytes on mid-game 19x19 parents - are never written.
template <int R, bool PACKED = false, bool COMPACT = false>
__global__ __launch_bounds__(kWave, GG_LB_CH3) void k_children3(const uint8_t *__restrict__ states,
                                                        uint8_t *__restrict__ children, int64_t B, int N,
                                                        uint32_t inv, int canonical, int chunks,
                                                        const int32_t *__restrict__ offsets = nullptr,
                                                        const int32_t *__restrict__ order = nullptr) {
  static_assert(!(PACKED && COMPACT), "compact children are byte planes");
  constexpr int RS = Cfg<R>::kRowStride;
  constexpr int RV = (R + 3) / 4;
    // (an 8 KB window: with the 16 KB one of rounds 2 - 3 the workgroup's LDS was 11 200 B, 14 waves per CU instead of 16)
  constexpr uint32_t kRingWords = 256, kRingBits = 32 * kRingWords, kBlk = 1024;   // 8 blocks of 1 KB output
  constexpr int kScWords = kWave * RS > 2 * Cfg<R>::kIoBytes / 4 ? kWave * RS : 2 * Cfg<R>::kIoBytes / 4;
  __shared__ __attribute__((aligned(16))) uint32_t sc[kScWords];   // flood results; the parent is staged here first
  __shared__ __attribute__((aligned(16))) uint32_t planes[4 * 32]; // rows: mover, opponent, both bit-reversed
  __shared__ __attribute__((aligned(16))) uint32_t ring[kRingWords];  // output bit-stream window (1 bit per output byte)
  __shared__ uint16_t elist[R * R + 2];                            // action index of the t-th empty point
  __shared__ uint16_t alist[R * R + 2];                            // index t of the u-th empty point NEXT TO A STONE
  __shared__ uint2 lut[256];
  const Half hf = make_half(threadIdx.x, N, inv);
  load_spread_lut(lut, hf.lane);
  const int S = 6 * hf.P;
  const int A = hf.P + 1;
  uint8_t *io = reinterpret_cast<uint8_t *>(sc) + hf.h * Cfg<R>::kIoBytes;
  for (int i = hf.lane; i < (int)kRingWords; i += kWave) ring[i] = 0;   // every flushed block is zeroed again
  const int per = (A + chunks - 1) / chunks;
  for (int64_t w = blockIdx.x; w < B * chunks; w += gridDim.x) {
    // COMPACT: the work of a parent grows with the children it keeps (57 late in a game, 342 early): in index order a launch
    // of two rounds of waves ends with a few heavy parents on an emptying machine (56 % mean occupancy measured).  `order`
    // (k_children_order_scan: the parents by falling count) hands the heavy ones out first - longest processing time first.
    const int64_t wp = w / chunks;
    const int64_t b = (COMPACT && order) ? (int64_t)order[wp] : wp;
    const int ch = (int)(w - wp * chunks);
    const uint8_t *gi = states + b * (int64_t)S;
    uint8_t *gc = COMPACT ? children + (int64_t)offsets[b] * S : children + b * A * (int64_t)S;
    const int W = 3 * N + 1;
    uint32_t *gcp = reinterpret_cast<uint32_t *>(children) + b * A * (int64_t)W;   // PACKED: this parent's child slots
    uint32_t flags, black, white, invd;   // flags: bit 0 turn, bit 2 passed, bit 3 done
    if (PACKED) {
      uint32_t fw;
      load_packed_h(reinterpret_cast<const uint32_t *>(states) + b * (int64_t)W, N, hf, black, white, invd, fw);
      flags = (fw & 1u) | ((fw & 6u) << 1);
      WAVE_SYNC();
    } else {
      PairRegs<R> pr;   // (flags and planes in flight together; both halves hold the same parent)
      bool no_tables = true;
      pair_issue<R>(pr, gi, 4 * hf.P, hf, true);
      const uint32_t mi = pair_commit<R>(pr, gi, 4 * hf.P, io, hf, nullptr, nullptr, no_tables, flags);
      black = plane_to_row<R>(io + mi, N, hf.hl);
      white = plane_to_row<R>(io + mi + hf.P, N, hf.hl);
      invd = plane_to_row<R>(io + mi + 3 * hf.P, N, hf.hl);
    }
    const int pl = flags & 1u;
    const uint32_t mine = pl ? white : black, opp = pl ? black : white;
    const uint32_t e = hf.full_l1 & ~(mine | opp);
    const int a0 = ch * per, a1 = min(A, a0 + per);
    if (a0 >= A) continue;         // trailing chunk with nothing in it
    const int p1 = min(a1, hf.P);  // points of the chunk: [a0, p1); the pass slot is in the chunk iff a1 == A
    const int base = hf.hl * N;
    const int lo = max(0, min(N, a0 - base)), hi = max(0, min(N, p1 - base));
    const uint32_t below_lo = (1u << lo) - 1u, below_hi = (1u << hi) - 1u;
    // empty points: total, list, and the index range [t0, t1) of those inside the chunk
    const uint32_t ecnt = (uint32_t)__popc(e);
    const uint32_t eincl = half_scan(ecnt);
    const int t0 = __builtin_amdgcn_readlane((int)half_scan((uint32_t)__popc(e & below_lo)), 31);
    const int t1 = __builtin_amdgcn_readlane((int)half_scan((uint32_t)__popc(e & below_hi)), 31);
    // Only an empty point NEXT TO A STONE can be a liberty, capture anything or join a group: the floods (pass 1 and
    // pass 2) run over those points only; a child on any other empty point is the parent plus a lone stone with >= 2
    // liberties.  Early-game parents (350 empty points, a few dozen next to stones) need 1-2 flood batches instead of 11.
    const uint32_t stn = mine | opp;
    const uint32_t ea = e & (B3(shl1(stn), stn >> 1, dpp0<0x138>(stn), T_OR3) | dpp0<0x130>(stn));
    const uint32_t acnt = (uint32_t)__popc(ea);
    const uint32_t aincl = half_scan(acnt);
    const int EA = __builtin_amdgcn_readlane((int)aincl, 31);
    const int u0 = __builtin_amdgcn_readlane((int)half_scan((uint32_t)__popc(ea & below_lo)), 31);   // ... inside the chunk: [u0, u1)
    const int u1 = __builtin_amdgcn_readlane((int)half_scan((uint32_t)__popc(ea & below_hi)), 31);
    // COMPACT: the actions valid_moves() keeps, their per-row prefix, and the ranks [k0, k1) of the chunk's first / last slot
    const uint32_t keep = !COMPACT ? 0u : ((flags & 8u) ? hf.full_l1 : hf.full_l1 & ~invd);
    const uint32_t kcnt = (uint32_t)__popc(keep);
    const uint32_t kincl = COMPACT ? half_scan(kcnt) : 0u;
    const int kall = COMPACT ? __builtin_amdgcn_readlane((int)kincl, 31) : 0;          // the pass's rank
    const int k0 = COMPACT ? __builtin_amdgcn_readlane((int)half_scan((uint32_t)__popc(keep & below_lo)), 31) : a0;
    const int k1 = COMPACT ? __builtin_amdgcn_readlane((int)half_scan((uint32_t)__popc(keep & below_hi)), 31) + (a1 == A ? 1 : 0) : a1;
    WAVE_SYNC();  // staging buffer read out
    if (hf.h == 0) {
      planes[hf.hl] = mine;
      planes[32 + hf.hl] = opp;
      planes[64 + hf.hl] = __brev(mine);
      planes[96 + hf.hl] = __brev(opp);
      if (hf.hl < N) {
        const uint32_t off = eincl - ecnt, aoff = aincl - acnt;
#pragma unroll
        for (int c = 0; c < R; ++c)
          if (c < N && ((e >> c) & 1u)) {
            const uint32_t t = off + (uint32_t)__popc(e & ((1u << c) - 1u));
            elist[t] = (uint16_t)(base + c);
            if ((ea >> c) & 1u) alist[aoff + (uint32_t)__popc(ea & ((1u << c) - 1u))] = (uint16_t)t;
          }
      }
    }
    WAVE_SYNC();

    // lane (h, l): flood colour h from the stones adjacent to the (ubase + l)-th empty point next to a stone; result ->
    // sc[lane][row].  Returns the lane's point (action index), -1 if there is none.
    auto flood_batch = [&](int ubase) -> int {
      uint32_t m[R], mrev[R], f[R];
      const int u = ubase + hf.hl;
      const bool have = u < EA;
      const int x = have ? (int)elist[alist[u]] : 0;
      int xr, xc;
      split_action(x, N, hf.inv, xr, xc);
      const uint32_t bit = have ? (1u << xc) : 0u, hb = (bit << 1) | (bit >> 1);
      {
        uint32_t mt[RV * 4], rt[RV * 4];
        const uint4 *pm = reinterpret_cast<const uint4 *>(planes + (hf.h ? 32 : 0));
        const uint4 *pr = reinterpret_cast<const uint4 *>(planes + 64 + (hf.h ? 32 : 0));
#pragma unroll
        for (int i = 0; i < RV; ++i) {
          const uint4 a = pm[i], c = pr[i];
          mt[4 * i] = a.x; mt[4 * i + 1] = a.y; mt[4 * i + 2] = a.z; mt[4 * i + 3] = a.w;
          rt[4 * i] = c.x; rt[4 * i + 1] = c.y; rt[4 * i + 2] = c.z; rt[4 * i + 3] = c.w;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          m[r] = mt[r];
          mrev[r] = rt[r];
          const uint32_t sd = (r == xr) ? hb : ((r == xr - 1 || r == xr + 1) ? bit : 0u);
          f[r] = m[r] & sd;
        }
      }
      WAVE_SYNC();  // earlier readers of sc are done
      flood2_serial<R>(m, mrev, f, sc + hf.lane * RS);
      WAVE_SYNC();
      return have ? x : -1;
    };

    // pass 1: exact liberty counts (saturated at 3) - half 0 counts for the mover's stones, half 1 for the opponent's
    uint32_t ge1 = 0, ge2 = 0, ge3 = 0;
    const int nbatch = (EA + 31) >> 5;
#pragma unroll 1
    for (int i = nbatch - 1; i >= 0; --i) {   // last batch first: batch 0 is the one still in sc when pass 2 starts
      flood_batch(32 * i);
      const int cnt = min(32, EA - 32 * i);
      if (hf.hl < R) {
        const uint32_t *col = sc + (32 * hf.h) * RS + hf.hl;
#pragma unroll 4
        for (int t = 0; t < cnt; ++t) {
          const uint32_t x = col[t * RS];
          ge3 = B3(ge2, x, ge3, T_ANDOR);
          ge2 = B3(ge1, x, ge2, T_ANDOR);
          ge1 |= x;
        }
      }
    }
    const uint32_t o2 = (uint32_t)__shfl((int)ge2, hf.lane ^ 32), o3 = (uint32_t)__shfl((int)ge3, hf.lane ^ 32);
    const uint32_t Mm = hf.h ? o2 : ge2;    // mover's stones with >= 2 liberties
    const uint32_t Mo = hf.h ? ge2 : o2;    // opponent's stones with >= 2 liberties
    const uint32_t T3o = hf.h ? ge3 : o3;   // ... with >= 3

    // Streaming emitter.  The slots [a0, a1) of this parent are one contiguous byte range; it is written strictly in
    // address order as ALIGNED 1 KB blocks (64 lanes x 16 B, the store pattern that reaches memset-like bandwidth:
    // tools/ubench/write_patterns.hip).  `ring` is a sliding window of the output as a bit-string (bit i = byte
    // origin[i]); a child ORs its 6 N^2 bits into it, a block leaves through the spread table once every child
    // overlapping it has been emitted, and the all-zero slots of the illegal actions cost nothing but zero stores.
    uint8_t *const cstart = gc + (int64_t)k0 * S;
    const uint32_t start_bit = (uint32_t)((uintptr_t)cstart & (kBlk - 1u));
    uint8_t *const origin = cstart - start_bit;
    const uint32_t end_bit = start_bit + (uint32_t)(k1 - k0) * (uint32_t)S;
    uint32_t sbase = 0, dirty_end = 0;   // wave-uniform: window start (multiple of kBlk); no bit set at or above dirty_end
    auto flush_until = [&](uint32_t target) {
      // four whole blocks per round (two adjacent slots are 4.2 blocks): the four ring reads, then the eight table reads,
      // then the four stores - two dependent LDS round trips per 4 KB instead of eight and a quarter of the loop's scalar
      // bookkeeping (round 4, with the 8 KB window: 1.23 / 1.17 / 1.14 -> 1.13 / 1.10 / 1.15 ms per 8 192 early- / mid- /
      // late-game parents)
#pragma unroll 1
      while (sbase + 4u * kBlk <= target && (sbase != 0u || start_bit == 0u) && sbase + 4u * kBlk <= end_bit) {
        uint8_t *dst = origin + sbase + 16u * (uint32_t)hf.lane;
        if (sbase < dirty_end) {
          const uint8_t *rb = reinterpret_cast<const uint8_t *>(ring);
          uint32_t hw[4];
#pragma unroll
          for (int k = 0; k < 4; ++k)
            hw[k] = *reinterpret_cast<const uint16_t *>(rb + (((sbase >> 3) + 128u * (uint32_t)k) & (4u * kRingWords - 1u)) + 2 * hf.lane);
          V16a o[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint2 lo = lut[hw[k] & 255u], hi = lut[hw[k] >> 8];
            o[k].w[0] = lo.x; o[k].w[1] = lo.y; o[k].w[2] = hi.x; o[k].w[3] = hi.y;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) *reinterpret_cast<V16a *>(dst + (uint32_t)k * kBlk) = o[k];
          asm volatile("" ::: "memory");
          ring[((sbase >> 5) + (uint32_t)hf.lane) & (kRingWords - 1u)] = 0;
          ring[((sbase >> 5) + 64u + (uint32_t)hf.lane) & (kRingWords - 1u)] = 0;
        } else {
          const V16a z = {{0u, 0u, 0u, 0u}};
#pragma unroll
          for (int k = 0; k < 4; ++k) *reinterpret_cast<V16a *>(dst + (uint32_t)k * kBlk) = z;
        }
        sbase += 4u * kBlk;
      }
#pragma unroll 1
      while (sbase < target) {
        const uint32_t off = sbase + 16u * (uint32_t)hf.lane;   // this lane's vector = bytes [off, off + 16)
        const bool dirty = sbase < dirty_end;
        V16a o = {{0u, 0u, 0u, 0u}};
        if (dirty) {
          const uint8_t *bb = reinterpret_cast<const uint8_t *>(ring) + ((sbase >> 3) & (4u * kRingWords - 1u));
          const uint2 lo = lut[bb[2 * hf.lane]], hi = lut[bb[2 * hf.lane + 1]];
          o.w[0] = lo.x; o.w[1] = lo.y; o.w[2] = hi.x; o.w[3] = hi.y;
        }
        if (off >= start_bit && off + 16u <= end_bit) {
          *reinterpret_cast<V16a *>(origin + off) = o;   // (a nontemporal store measures the same: 4.9 TB/s either way)
        } else if (off + 16u > start_bit && off < end_bit) {   // the ragged vector at either end of the chunk: bytes
#pragma unroll
          for (int t = 0; t < 16; ++t)
            if (off + t >= start_bit && off + t < end_bit) origin[off + t] = (uint8_t)(o.w[t >> 2] >> (8 * (t & 3)));
        }
        asm volatile("" ::: "memory");
        if (dirty && hf.lane < 32) ring[((sbase >> 5) & (kRingWords - 1u)) + hf.lane] = 0;
        sbase += kBlk;
      }
    };
    auto or_bits = [&](uint32_t p, uint32_t b0, uint32_t b1, uint32_t b3, uint32_t turn, uint32_t passed,
                       uint32_t done, bool wr) {
      if (wr && hf.hl < hf.N) {
        const uint32_t rows[6] = {b0, b1, turn ? hf.full_l1 : 0u, b3, passed ? hf.full_l1 : 0u, done ? hf.full_l1 : 0u};
        const uint32_t q0 = p + (uint32_t)(hf.hl * hf.N);
#pragma unroll
        for (int pn = 0; pn < 6; ++pn) {
          if (rows[pn]) {
            const uint32_t q = q0 + (uint32_t)(pn * hf.P);
            const uint64_t x = (uint64_t)rows[pn] << (q & 31u);
            const uint32_t wi = (q >> 5) & (kRingWords - 1u);
            atomicOr(&ring[wi], (uint32_t)x);
            if ((uint32_t)(x >> 32)) atomicOr(&ring[(wi + 1u) & (kRingWords - 1u)], (uint32_t)(x >> 32));
          }
        }
      }
    };
    if (PACKED) {
      // packed slots: zero the chunk's slots with coalesced stores, wait for them, then overwrite the legal ones
      uint32_t *z = gcp + (int64_t)a0 * W;
      const int nz = (a1 - a0) * W;
      for (int i = hf.lane; i < nz; i += kWave) z[i] = 0u;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      WAVE_SYNC();
    }
    // one child per half: `a` = its action, `tj` = flood lane of its point inside the current batch (-1: pass, -2: an
    // empty point with no stone next to it - no group is touched, nothing to look up)
    auto child = [&](int a, int tj, bool on, int slot) {   // slot: the child's place among the parent's slots (a, or a's rank)
      const bool is_pass = tj == -1;
      uint32_t nmine = mine, nopp = opp, invalid;
      if (__ballot(on && !is_pass) == 0) {
        invalid = invalid_from2(opp, mine, Mo, Mm, hf);
      } else {
        const int tl = tj < 0 ? 0 : tj;
        uint32_t Rm = 0, Ro = 0;
        if (hf.hl < R && tj >= 0) {
          Rm = sc[tl * RS + hf.hl];
          Ro = sc[(32 + tl) * RS + hf.hl];
        }
        int ra, ca;
        split_action(is_pass ? 0 : a, N, hf.inv, ra, ca);
        const uint32_t bit = is_pass ? 0u : (1u << ca);
        const uint32_t qrow = hf.hl == ra ? bit : 0u;
        uint32_t nbm = hf.hl == ra ? ((bit << 1) | (bit >> 1)) : ((hf.hl == ra - 1 || hf.hl == ra + 1) ? bit : 0u);
        nbm &= hf.full_l1;
        const bool boxed = half_of(__ballot((nbm & ~opp) != 0), hf.h) == 0;
        const uint32_t cap = Ro & ~Mo;               // adjacent opponent groups whose only liberty was q
        nopp = opp & ~cap;
        nmine = mine | qrow;
        const uint32_t G = Rm | qrow;
        const uint32_t e2 = hf.full_l1 & ~(nmine | nopp);
        const uint32_t dil = B3(shl1(G), G >> 1, dpp0<0x138>(G), T_OR3) | dpp0<0x130>(G);
        const uint32_t libs = dil & e2;
        const uint32_t nz = half_of(__ballot(libs != 0), hf.h);
        const uint32_t many = half_of(__ballot(__popc(libs) > 1), hf.h);
        const bool multiG = ((nz & (nz - 1u)) | many) != 0;
        uint32_t Mm2 = (Mm & ~Rm) | (multiG ? G : 0u);
        if (__ballot(cap != 0)) {
          // gogame.py:72-75: ko iff exactly one stone died and the new stone is boxed in
          const uint32_t dm = half_of(__ballot(cap != 0), hf.h);
          const uint32_t manyc = half_of(__ballot(__popc(cap) > 1), hf.h);
          int ko_r = -1, ko_c = 0;
          if (dm && boxed && manyc == 0 && (dm & (dm - 1u)) == 0) {
            ko_r = __ffs(dm) - 1;
            ko_c = __ffs(__shfl(cap, (hf.lane & 32) + ko_r)) - 1;
          }
          // mover's groups in atari next to a captured group (and not merged into G) now have >= 2 liberties
          const uint32_t atari_m = mine & ~Mm;
          const uint32_t dcap = B3(shl1(cap), cap >> 1, dpp0<0x138>(cap), T_OR3) | dpp0<0x130>(cap);
          uint32_t f = dcap & atari_m & ~G;
          if (__ballot(f != 0)) {
#pragma unroll 1
            for (int it = 0; it < R * R; ++it) {
              const uint32_t grow = B3(shl1(f), f >> 1, dpp0<0x138>(f), T_OR3) | dpp0<0x130>(f);
              const uint32_t g = B3(grow, atari_m, f, T_ANDOR);
              const bool chg = g != f;
              f = g;
              if (__ballot(chg) == 0) break;
            }
            Mm2 |= f;
          }
          const uint32_t Mo2 = (Mo & ~Ro) | (Ro & T3o);
          invalid = invalid_from2(nopp, nmine, Mo2, Mm2, hf);
          if (hf.hl == ko_r) invalid |= 1u << ko_c;
        } else {
          const uint32_t Mo2 = (Mo & ~Ro) | (Ro & T3o);
          invalid = invalid_from2(nopp, nmine, Mo2, Mm2, hf);
        }
      }
      uint32_t nb = pl ? nopp : nmine, nw = pl ? nmine : nopp;
      const uint32_t passed = is_pass ? 1 : 0;
      const uint32_t done = ((flags & 8u) || (is_pass && (flags & 4u))) ? 1 : 0;
      int nturn = 1 - pl;
      if (canonical && nturn == 1) {
        const uint32_t t = nb; nb = nw; nw = t;
        nturn = 0;
      }
      if (PACKED) {
        store_packed_h(gcp + (int64_t)a * W, N, hf, nb, nw, invalid, (uint32_t)nturn, passed, done, on);
        return;
      }
      // both children are emitted in slot order; the window is advanced up to the block the next child starts in
      const uint32_t pbit = start_bit + (uint32_t)(slot - k0) * (uint32_t)S;
      const uint32_t pA = (uint32_t)__builtin_amdgcn_readlane((int)pbit, 0), pB = (uint32_t)__builtin_amdgcn_readlane((int)pbit, 32);
      const bool onB = (__ballot(on) >> 32) != 0;
      WAVE_SYNC();
      flush_until(pA & ~(kBlk - 1u));
      WAVE_SYNC();
      if (!onB || pB + (uint32_t)S <= sbase + kRingBits) {
        or_bits(pbit, nb, nw, invalid, (uint32_t)nturn, passed, done, on);
        dirty_end = onB ? pB + (uint32_t)S : pA + (uint32_t)S;
      } else {
        or_bits(pbit, nb, nw, invalid, (uint32_t)nturn, passed, done, on && hf.h == 0);
        dirty_end = pA + (uint32_t)S;
        WAVE_SYNC();
        flush_until(pB & ~(kBlk - 1u));
        WAVE_SYNC();
        or_bits(pbit, nb, nw, invalid, (uint32_t)nturn, passed, done, on && hf.h == 1);
        dirty_end = pB + (uint32_t)S;
      }
    };

    // pass 2: the chunk's empty points in slot order.  Window i = the i-th batch of 32 stone-adjacent points of the chunk
    // (one flood batch) together with every free point up to the next window; its points are taken 32 at a time (one per
    // lane of the lower half), two legal children per L1 pass.
    {
      const int nwin = u1 > u0 ? (u1 - u0 + 31) >> 5 : 1;
#pragma unroll 1
      for (int wi = 0; wi < nwin; ++wi) {
        const int ub = u0 + 32 * wi;
        if (ub < u1 && ub != 0) flood_batch(ub);   // (pass 1 ended with batch 0: it is still in sc)
        const int tlo = wi == 0 ? t0 : (int)alist[ub];
        const int thi = (wi + 1 < nwin) ? (int)alist[ub + 32] : t1;
#pragma unroll 1
        for (int tb = tlo; tb < thi; tb += 32) {
          const int t = tb + hf.hl;
          const int x = t < thi ? (int)elist[t] : -1;
          int xr, xc;
          split_action(x < 0 ? 0 : x, N, hf.inv, xr, xc);
          const uint32_t irow = __shfl(invd, xr);   // every lane executes the exchanges (a masked-off source lane reads as 0)
          const uint32_t arow = __shfl(ea, xr);
          const uint32_t apre = __shfl(aincl - acnt, xr);
          const uint32_t krow = __shfl(keep, xr), kpre = __shfl(kincl - kcnt, xr);
          const int rk = COMPACT ? (int)(kpre + (uint32_t)__popc(krow & ((1u << xc) - 1u))) : x;
          const bool legal = x >= 0 && ((irow >> xc) & 1u) == 0;
          // flood lane of the point inside the window's batch, or -2 for a free point
          const int fl = ((arow >> xc) & 1u) ? (int)(apre + (uint32_t)__popc(arow & ((1u << xc) - 1u))) - ub : -2;
          uint32_t lm = (uint32_t)__ballot(legal);   // low half: lanes 0-31 carry the 32 points of the round
#pragma unroll 1
          while (lm) {
            const int j0 = __ffs(lm) - 1;
            lm &= lm - 1u;
            const int j1 = lm ? __ffs(lm) - 1 : -1;
            lm &= lm - 1u;
            const int js = hf.h ? j1 : j0;
            const bool on = js >= 0;
            const int src = on ? js : j0;
            const int a = __shfl(x, src);
            const int tj = __shfl(fl, src);
            child(a, tj, on, __shfl(rk, src));
          }
        }
      }
    }
    if (a1 == A) child(hf.P, -1, hf.h == 0, COMPACT ? kall : hf.P);
    if (!PACKED) {
      WAVE_SYNC();
      flush_until((end_bit + kBlk - 1u) & ~(kBlk - 1u));
      WAVE_SYNC();
    }
  }
}

// gogame.children(padded=False): how many children valid_moves() keeps per parent (gym_go/gogame.py:153-161, 176-179) - the
// points whose plane-3 byte is clear + the pass; every action once the game has ended.  One wave per parent at a time.
static __global__ void k_children_counts(const uint8_t *__restrict__ states, int32_t *__restrict__ counts, int64_t B, int N) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / kWave;
  const int64_t nwaves = (gridDim.x * (int64_t)blockDim.x) / kWave;
  const int P = N * N;
  for (int64_t b = wave; b < B; b += nwaves) {
    const uint8_t *g = states + b * (int64_t)(6 * P);
    int c = 0;
    for (int i = lane; i < P; i += kWave) c += (g[3 * P + i] & 1) == 0 ? 1 : 0;   // (bit 0, like plane_to_row: the expansion must keep exactly what is counted here)
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if (lane == 0) counts[b] = g[5 * P] ? P + 1 : c + 1;
  }
}

// The per-parent counts of a children batch -> what the compact expansion needs, in ONE workgroup of 1 024 threads and one
// launch (round 5 ran two: a counting sort whose bin scan was one thread walking 363 LDS words, 11.9 us, and a tiled scan
// with three barriers per 1 024 parents, 8.8 us - together 6 % of the compact call on 8 192 parents):
//   * order[0 .. B) (optional) = the parents by FALLING child count (a counting sort over the <= N*N+1 possible counts; ties in
//     any order: it only decides which work item a parent is, never what is written) - the launch order of the expansion;
//   * v[0 .. B) = the counts' exclusive prefix sums in place, v[B] = the total: every thread owns a contiguous run of parents
//     (serial inside the run, one scan of the 1 024 run totals across the workgroup).
static __global__ __launch_bounds__(1024) void k_children_order_scan(int32_t *__restrict__ v, int32_t *__restrict__ order,
                                                                     int64_t B, int A) {
  __shared__ int32_t bin[512];    // A <= 362
  __shared__ int32_t wsum[16];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  if (order) {
    if (tid < 512) bin[tid] = 0;
    __syncthreads();
    for (int64_t i = tid; i < B; i += 1024) atomicAdd(&bin[A - v[i]], 1);    // bin 0 = the largest count
    __syncthreads();
    {   // exclusive scan of the 512 bins: eight waves, then the wave totals
      const int32_t c = tid < 512 ? bin[tid] : 0;
      int32_t incl = c;
      for (int d = 1; d < kWave; d <<= 1) {
        const int32_t y = __shfl_up(incl, d);
        if (lane >= d) incl += y;
      }
      if (lane == kWave - 1) wsum[wv] = incl;
      __syncthreads();
      int32_t before = 0;
      for (int k = 0; k < wv; ++k) before += wsum[k];
      if (tid < 512) bin[tid] = before + incl - c;
      __syncthreads();
    }
    for (int64_t i = tid; i < B; i += 1024) order[atomicAdd(&bin[A - v[i]], 1)] = (int32_t)i;
    __syncthreads();   // (the counts are read above and overwritten below)
  }
  const int64_t run = (B + 1023) / 1024;
  if (run <= 16) {   // a children batch of up to 16 384 parents: one pass, two barriers
    const int64_t lo = (int64_t)tid * run, hi = lo + run < B ? lo + run : B;
    int32_t sum = 0;
    for (int64_t i = lo; i < hi; ++i) sum += v[i];
    int32_t incl = sum;
    for (int d = 1; d < kWave; d <<= 1) {
      const int32_t y = __shfl_up(incl, d);
      if (lane >= d) incl += y;
    }
    if (lane == kWave - 1) wsum[wv] = incl;
    __syncthreads();
    int32_t acc = incl - sum;
    for (int k = 0; k < wv; ++k) acc += wsum[k];
    for (int64_t i = lo; i < hi; ++i) { const int32_t x = v[i]; v[i] = acc; acc += x; }
    if (tid == 1023) v[B] = acc;   // (the last thread's run ends at B - or is empty, then acc is the total before it)
    return;
  }
  // larger batches: tiles of 1 024 consecutive parents (coalesced), the running total carried from tile to tile
  __shared__ int32_t carry_s;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < B; base += 1024) {
    const int64_t i = base + tid;
    const int32_t x = i < B ? v[i] : 0;
    int32_t incl = x;
    for (int d = 1; d < kWave; d <<= 1) {
      const int32_t y = __shfl_up(incl, d);
      if (lane >= d) incl += y;
    }
    if (lane == kWave - 1) wsum[wv] = incl;
    __syncthreads();
    int32_t before = carry_s;
    for (int k = 0; k < wv; ++k) before += wsum[k];
    if (i < B) v[i] = before + incl - x;
    __syncthreads();
    if (tid == 1023) carry_s = before + incl;
    __syncthreads();
  }
  if (tid == 0) v[B] = carry_s;
}

// state_utils.batch_compute_invalid_moves (gym_go/state_utils.py:86-156), two boards per wave: plane 3 recomputed from
// planes 0-2 (+ an optional ko point per game).
template <int R, bool FULLN = false>
__global__ __launch_bounds__(kWave, 4) void k_invalid_mask2(const uint8_t *__restrict__ states,
                                                            const int32_t *__restrict__ ko, uint8_t *__restrict__ mask,
                                                            int64_t B, int N, uint32_t inv, AgeSplit age) {
  if (FULLN) { N = R; inv = (65536u + R - 1u) / R; }   // N == R: compile-time constants (see k_env_step2)
  __shared__ __attribute__((aligned(16))) uint32_t lds[Lds2<R>::kTotal];
  const Half hf = make_half(threadIdx.x, N, inv);
  bool tables = false;
  const int S = 6 * hf.P;
  uint8_t *io = reinterpret_cast<uint8_t *>(lds) + hf.h * Cfg<R>::kIoBytes;
  const int64_t npairs = (B + 1) >> 1;
  const PairSpan span = pair_span(npairs, age);
  for (int64_t p = span.first; p < span.end; p += span.stride) {
    const bool on = 2 * p + hf.h < B;
    const int64_t b = on ? 2 * p + hf.h : B - 1;
    const uint8_t *gi = states + b * (int64_t)S;
    PairRegs<R> pr;
    pair_issue<R>(pr, gi, 2 * hf.P, hf, tables);
    const int k = ko ? ko[b] : -1;
    uint32_t flags;
    const uint32_t mi = pair_commit<R>(pr, gi, 2 * hf.P, io, hf, lds, nullptr, tables, flags);
    const uint32_t black = plane_to_row<R>(io + mi, N, hf.hl);
    const uint32_t white = plane_to_row<R>(io + mi + hf.P, N, hf.hl);
    const int nx = flags & 1u;  // side to move
    const uint32_t nxs = nx ? white : black, pls = nx ? black : white;
    uint32_t multi_nx, alive_nx, multi_pl;
    analyze2<R, false>(nxs, pls, hf.full_l1 & ~(black | white), hf, lds, multi_nx, alive_nx, multi_pl);
    uint32_t invalid = invalid_from2(nxs, pls, multi_nx, multi_pl, hf);
    if (k >= 0 && k < hf.P) {
      const int kr = (int)(((uint32_t)k * inv) >> 16), kc = k - kr * N;
      if (hf.hl == kr) invalid |= 1u << kc;
    }
    uint8_t *gm = mask + b * (int64_t)hf.P;
    WAVE_SYNC();
    row_to_plane<R>(io + ((uintptr_t)gm & 15u), invalid, N, hf.hl);
    WAVE_SYNC();
    stage_out_h(gm, hf.P, io, hf.hl, on);
  }
}

// ---------------------------------------------------------------- bit-packed state format
// One board = 3 N + 1 uint32: N row masks (bit c = column c) of plane 0 (black), plane 1 (white), plane 3 (invalid
// moves), then one flag word (bit 0 turn, bit 1 previous move was a pass, bit 2 game over).  19x19: 232 B instead of
// 2 166 B.  Both directions are pure streaming kernels (half a wave per board, the L1 row layout IS the format).
template <int R>
__global__ __launch_bounds__(kWave) void k_pack(const uint8_t *__restrict__ states, uint32_t *__restrict__ packed,
                                                int64_t B, int N) {
  __shared__ __attribute__((aligned(16))) uint8_t iobuf[2][Cfg<R>::kIoBytes];
  const Half hf = make_half(threadIdx.x, N, 0);
  const int S = 6 * hf.P, W = 3 * N + 1;
  uint8_t *io = iobuf[hf.h];
  const int64_t npairs = (B + 1) >> 1;
  for (int64_t p = blockIdx.x; p < npairs; p += gridDim.x) {
    const bool on = 2 * p + hf.h < B;
    const int64_t b = on ? 2 * p + hf.h : B - 1;
    const uint8_t *gs = states + b * (int64_t)S;
    const uint32_t flags = load_flags_h(gs, hf.P, 0, hf);
    WAVE_SYNC();
    const uint32_t mi = stage_in_h(gs, 4 * hf.P, io, hf.hl);
    WAVE_SYNC();
    const uint32_t black = plane_to_row<R>(io + mi, N, hf.hl);
    const uint32_t white = plane_to_row<R>(io + mi + hf.P, N, hf.hl);
    const uint32_t invalid = plane_to_row<R>(io + mi + 3 * hf.P, N, hf.hl);
    uint32_t *gp = packed + b * (int64_t)W;
    if (on && hf.hl < N) {
      gp[hf.hl] = black;
      gp[N + hf.hl] = white;
      gp[2 * N + hf.hl] = invalid;
    }
    if (on && hf.hl == 31) gp[3 * N] = (flags & 1u) | ((flags >> 1) & 6u);  // flags: bit0 turn, bit2 passed, bit3 done
  }
}

template <int R>
__global__ __launch_bounds__(kWave) void k_unpack(const uint32_t *__restrict__ packed, uint8_t *__restrict__ states,
                                                  int64_t B, int N, int planes) {   // planes: 3 packed, 5 tracked
  __shared__ __attribute__((aligned(16))) uint32_t lds[2 * (Cfg<R>::kIoBytes / 4)];
  __shared__ uint2 lut[256];
  const Half hf = make_half(threadIdx.x, N, 0);
  load_spread_lut(lut, hf.lane);
  const int S = 6 * hf.P, W = planes * N + 1;
  uint32_t *work = lds + hf.h * (Cfg<R>::kIoBytes / 4);
  const int64_t npairs = (B + 1) >> 1;
  for (int64_t p = blockIdx.x; p < npairs; p += gridDim.x) {
    const bool on = 2 * p + hf.h < B;
    const int64_t b = on ? 2 * p + hf.h : B - 1;
    const uint32_t *gp = packed + b * (int64_t)W;
    const uint32_t full = hf.full_l1;
    uint32_t black = 0, white = 0, invalid = 0;
    if (hf.hl < N) {
      black = gp[hf.hl] & full;
      white = gp[N + hf.hl] & full;
      invalid = gp[2 * N + hf.hl] & full;
    }
    const uint32_t fl = gp[planes * N];
    emit_store_h<R>(states + b * (int64_t)S, black, white, invalid, fl & 1u, (fl >> 1) & 1u, (fl >> 2) & 1u, hf, work, lut,
                    on);
  }
}

}  // namespace gg
