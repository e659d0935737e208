// gg_v2.h - kernel family v2 (default): TWO BOARDS PER WAVEFRONT, constant-weight liberty code.
#pragma once
#include "gg_common.h"

namespace gg {

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
// The flood variants, the 2-cycle op spelling (bitop3 / add-for-shift) and the staging helpers are in gg_common.h.
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
  static constexpr int kTotal = kCwt + (kCwClasses + 1) * 20;
};

// Liberty analysis of both boards of the wave (L1 in, L1 out; see analyze<R> for the single-board form).
template <int R, bool DUAL>
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
  if (DUAL) flood2_dual<R>(m, mrev, f, sc + hf.lane * RS);
  else flood2_serial<R>(m, mrev, f, sc + hf.lane * RS);
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
template <int R, bool DUAL>
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
    analyze2<R, DUAL>(opp, mine, e, hf, lds, multi_opp, alive_opp, multi_mine);
  } else {
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      uint32_t e = hf.full_l1 & ~(mine | opp);
      analyze2<R, DUAL>(opp, mine, e, hf, lds, multi_opp, alive_opp, multi_mine);
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
//      two words of bs[], 4 cells -> 4 bytes by a 24-bit multiply;
//   3. vectors that lie inside the board go straight from registers to HBM (global_store_dwordx4); the (at
//      most two) ragged ones are parked in LDS and leave in ONE global_store_byte instruction.
// `work` = the half's LDS staging area (>= 96 + 8 words).
template <int R>
__device__ __forceinline__ void emit_store_h(uint8_t *g, uint32_t black, uint32_t white, uint32_t invalid,
                                             uint32_t turn, uint32_t passed, uint32_t done, const Half &hf,
                                             uint32_t *work, bool wr) {
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
        // 4 cells -> 4 bytes: (bits * 0x204081) & 0x01010101 (bit i lands at bit 8 i; no colliding partial products)
        o.w[0] = __umul24(b16 & 15u, 0x204081u) & 0x01010101u;
        o.w[1] = __umul24((b16 >> 4) & 15u, 0x204081u) & 0x01010101u;
        o.w[2] = __umul24((b16 >> 8) & 15u, 0x204081u) & 0x01010101u;
        o.w[3] = __umul24((b16 >> 12) & 15u, 0x204081u) & 0x01010101u;
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
__global__ __launch_bounds__(kWave, 3) void k_next_states2(const uint8_t *__restrict__ in,
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
    uint32_t invalid = step_core2<R, false>(mine, opp, illegal ? hf.P : a, hf, lds, 0u, false, atari_unused);
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
                    reinterpret_cast<uint32_t *>(io), on && !illegal);
    if (illegal) copy_row_h(gi, go, S, hf.hl, on);  // rare: the row passes through unchanged
    if (status && on && hf.hl == 0) status[b] = illegal ? GG_STATUS_ILLEGAL : GG_STATUS_OK;
  }
}

// PERPLY = instantiation for 1-2 plies per launch: the board I/O dominates there and the kernel runs best spill-free
// at 3 waves per SIMD; the fused instantiation keeps its hot ply loop spill-free at 4 waves per SIMD.
template <int R, bool PERPLY>
__global__ __launch_bounds__(kWave, PERPLY ? 3 : 4) void k_rollout2(uint8_t *__restrict__ states, uint64_t *__restrict__ rng,
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
      uint32_t ninv = step_core2<R, false>(mine, opp, live ? a : hf.P, hf, lds, atari, have_atari, natari);
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
                      reinterpret_cast<uint32_t *>(io), on && played != 0);
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
      uint32_t invalid = step_core2<R, true>(mine, opp, a, hf, lds, 0u, false, atari_unused);
      uint32_t nb = pl ? opp : mine, nw = pl ? mine : opp;
      uint32_t passed = is_pass ? 1 : 0;
      uint32_t done = ((flags & 8u) || (is_pass && (flags & 4u))) ? 1 : 0;
      int nturn = 1 - pl;
      if (canonical && nturn == 1) {
        uint32_t t = nb; nb = nw; nw = t;
        nturn = 0;
      }
      uint8_t *go = gc + (int64_t)a * S;
      emit_store_h<R>(go, nb, nw, invalid, (uint32_t)nturn, passed, done, hf, reinterpret_cast<uint32_t *>(io), on);
      for (int t = 0; t < q && az < p1; ++t, az += 2) zero_step(az);
    }
    for (; az < p1; az += 2) zero_step(az);
  }
}

// state_utils.batch_compute_invalid_moves (gym_go/state_utils.py:86-156), two boards per wave: plane 3 recomputed from
// planes 0-2 (+ an optional ko point per game).
template <int R>
__global__ __launch_bounds__(kWave, 4) void k_invalid_mask2(const uint8_t *__restrict__ states,
                                                            const int32_t *__restrict__ ko, uint8_t *__restrict__ mask,
                                                            int64_t B, int N, uint32_t inv) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[Lds2<R>::kTotal];
  const Half hf = make_half(threadIdx.x, N, inv);
  load_cw_table<R>(lds, hf.lane);
  const int S = 6 * hf.P;
  uint8_t *io = reinterpret_cast<uint8_t *>(lds) + hf.h * Cfg<R>::kIoBytes;
  const int64_t npairs = (B + 1) >> 1;
  for (int64_t p = blockIdx.x; p < npairs; p += gridDim.x) {
    const bool on = 2 * p + hf.h < B;
    const int64_t b = on ? 2 * p + hf.h : B - 1;
    const uint8_t *gi = states + b * (int64_t)S;
    const uint32_t flags = load_flags_h(gi, hf.P, 0, hf);
    const int k = ko ? ko[b] : -1;
    WAVE_SYNC();
    const uint32_t mi = stage_in_h(gi, 2 * hf.P, io, hf.hl);
    WAVE_SYNC();
    const uint32_t black = plane_to_row<R>(io + mi, N, hf.hl);
    const uint32_t white = plane_to_row<R>(io + mi + hf.P, N, hf.hl);
    const int nx = flags & 1u;  // side to move
    const uint32_t nxs = nx ? white : black, pls = nx ? black : white;
    uint32_t multi_nx, alive_nx, multi_pl;
    analyze2<R, true>(nxs, pls, hf.full_l1 & ~(black | white), hf, lds, multi_nx, alive_nx, multi_pl);
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

// gogame.areas (gym_go/gogame.py:275-300), two boards per wave: in each half lane 0 floods the empty points from
// those touching black, lane 1 from those touching white; a region reached by exactly one colour belongs to it.
template <int R>
__global__ __launch_bounds__(kWave) void k_areas2(const uint8_t *__restrict__ states, int32_t *__restrict__ black_area,
                                                  int32_t *__restrict__ white_area, int64_t B, int N) {
  constexpr int RS = Cfg<R>::kRowStride;
  constexpr int RV = (R + 3) / 4;
  __shared__ __attribute__((aligned(16))) uint32_t lds[Lds2<R>::kTotal];
  const Half hf = make_half(threadIdx.x, N, 0);
  const int S = 6 * hf.P;
  uint8_t *io = reinterpret_cast<uint8_t *>(lds) + hf.h * Cfg<R>::kIoBytes;
  uint32_t *sc = lds;
  uint32_t *my5 = lds + Lds2<R>::kRows5 + hf.h * 160;
  const int64_t npairs = (B + 1) >> 1;
  for (int64_t p = blockIdx.x; p < npairs; p += gridDim.x) {
    const bool on = 2 * p + hf.h < B;
    const int64_t b = on ? 2 * p + hf.h : B - 1;
    WAVE_SYNC();
    const uint32_t mi = stage_in_h(states + b * (int64_t)S, 2 * hf.P, io, hf.hl);
    WAVE_SYNC();
    const uint32_t black = plane_to_row<R>(io + mi, N, hf.hl);
    const uint32_t white = plane_to_row<R>(io + mi + hf.P, N, hf.hl);
    const uint32_t e = hf.full_l1 & ~(black | white);
    WAVE_SYNC();
    my5[hf.hl] = e;
    my5[32 + hf.hl] = __brev(e);
    my5[64 + hf.hl] = black;
    my5[96 + hf.hl] = white;
    WAVE_SYNC();
    uint32_t m[R], mrev[R], f[R];
    {
      uint32_t mt[RV * 4], mr[RV * 4], src[RV * 4 + 1];
      const uint4 *pm = reinterpret_cast<const uint4 *>(my5);
      const uint4 *pr = reinterpret_cast<const uint4 *>(my5 + 32);
      const uint4 *ps = reinterpret_cast<const uint4 *>(my5 + (hf.hl == 1 ? 96 : 64));
#pragma unroll
      for (int i = 0; i < RV; ++i) {
        uint4 a = pm[i], c = pr[i], d = ps[i];
        mt[4 * i] = a.x; mt[4 * i + 1] = a.y; mt[4 * i + 2] = a.z; mt[4 * i + 3] = a.w;
        mr[4 * i] = c.x; mr[4 * i + 1] = c.y; mr[4 * i + 2] = c.z; mr[4 * i + 3] = c.w;
        src[4 * i] = d.x; src[4 * i + 1] = d.y; src[4 * i + 2] = d.z; src[4 * i + 3] = d.w;
      }
      src[RV * 4] = 0;
      const uint32_t use = hf.hl < 2 ? 0xFFFFFFFFu : 0u;  // only lanes 0 / 1 of a half carry a flood
#pragma unroll
      for (int r = 0; r < R; ++r) {
        m[r] = mt[r];
        mrev[r] = mr[r];
        uint32_t x = r > 0 ? B3(shl1(src[r]), src[r] >> 1, src[r - 1], T_OR3) : (shl1(src[r]) | (src[r] >> 1));
        f[r] = B3(m[r], x, r < R - 1 ? src[r + 1] : 0u, T_AND_OR2) & use;
      }
    }
    WAVE_SYNC();
    flood2_dual<R>(m, mrev, f, sc + hf.lane * RS);
    WAVE_SYNC();
    uint32_t cb = 0, cw = 0;
    if (hf.hl < R) {
      const uint32_t fb = sc[(hf.h * 32) * RS + hf.hl], fw = sc[(hf.h * 32 + 1) * RS + hf.hl];
      cb = (uint32_t)__popc(black) + (uint32_t)__popc(fb & ~fw);
      cw = (uint32_t)__popc(white) + (uint32_t)__popc(fw & ~fb);
    }
    cb = half_scan(cb);
    cw = half_scan(cw);
    if (on && hf.hl == 31) {
      black_area[b] = (int32_t)cb;
      white_area[b] = (int32_t)cw;
    }
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
                                                  int64_t B, int N) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[2 * (Cfg<R>::kIoBytes / 4)];
  const Half hf = make_half(threadIdx.x, N, 0);
  const int S = 6 * hf.P, W = 3 * N + 1;
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
    const uint32_t fl = gp[3 * N];
    emit_store_h<R>(states + b * (int64_t)S, black, white, invalid, fl & 1u, (fl >> 1) & 1u, (fl >> 2) & 1u, hf, work,
                    on);
  }
}

}  // namespace gg
