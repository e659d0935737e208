// gg_v5.h - the multi-ply kernel for batches that fill the machine: THIRTY-TWO BOARDS PER WAVEFRONT (one pair of lanes per
// board), the floods of a ply compacted into a job list.
#pragma once
#include "gg_v4.h"

namespace gg {

// ===================================================================== v5: 32 boards per wave, flood jobs
// k_rollout4 (gg_v4.h) gives every board a quad of lanes: lane t floods whatever q's neighbour in direction t holds, and
// 39 % of those flood lanes carry a flood (1.56 per board and ply) - the flood batch, the liberty count and the seed set-up,
// 54 % of a ply's instructions, cost the same whatever their lanes carry.  Here a board is a PAIR of lanes (lane t owns
// the RPL = ceil(R / 2) adjacent rows RPL t ..), a wave holds 32 boards, and a ply is
//   1.  sampling, as in k_rollout4 (the pair scan is one DPP swap, the row of the k-th point a search tree over the prefix
//       counts);
//   2a. the board's own lanes look at q's four neighbours (two directions each) and post the floods the ply needs as JOBS:
//       one per opponent stone next to q, and ONE for the mover's group G when q has a friendly neighbour (k_rollout4 runs
//       that flood in every friendly lane); the slots come from two ballots (v_mbcnt prefix), the descriptors go to LDS;
//   2b. lane L runs job L - 32 boards x 1.56 = 50 jobs in 64 lanes (a second batch when a ply posts more than 64: 0.7 %) -
//       with the fill in registers from the seed to the liberty count.  The batch's loop ends as soon as every flood of a G
//       is closed as far as stones OUTSIDE M go (the rows of M come out of the board lanes' registers by ds_bpermute): the
//       minimum of three sweeps on 98 % of the batches; an opponent group whose part found so far has two liberties is
//       settled, the few lanes left with fewer flood on.  What phase 3 needs leaves the lane at the end: G as a block of its
//       board, an opponent group that is captured or leaves M ORed into the board's collection block, a captured direction
//       and G's liberties ORed into the board's info word;
//   3.  class patch, captures, ko and the next mover's mask on the pair lanes, from ONE collection block per board: it
//       splits by M alone (captured = not in M: q was the only liberty).
// Instructions per board: the flood batch is shared by twice the boards, and so is everything in phases 1 and 3 that does not
// scale with the rows a lane holds (the draw, the k-th-bit search, the capture / ko logic, addresses): 47.6 VALU per env step
// (PMC) in the first version against k_rollout4's 73.2, fewer since.  65 536 games are 2 048 waves = TWO per SIMD (19.5 KB of
// LDS, up to 256 VGPRs per wave); a SIMD with two waves sustains one dependent VALU instruction per 1.77 ns against 1.60
// with four (tools/ubench/dep_chain.hip), which is why the kernel pays from 256 games per CU on and not below.
// Scope: drawn moves on full-size boards (N == R), byte planes or tracked boards - the fused rollout of big batches; every
// other form stays on k_rollout4.
constexpr int kNB5 = 32;
constexpr int kJobCap = 128;   // jobs of one ply: <= 4 per board (four opponent neighbours leave no friendly one)

template <int R>
struct Lds5 {
  // words per row block.  (13x13: 20 instead of the 16 of the other kernels - the lanes of a wave address 32 or 64 blocks at once,
  // and a stride of 16 words puts every fourth of them on the same LDS banks: 1.42 -> ms per launch of 65 536 games x 256 plies)
  static constexpr int RS = R == 13 ? 20 : Cfg<R>::kRowStride;
  static constexpr int RPL = (R + 1) / 2;                        // rows per lane in phases 1 and 3
  static_assert(2 * RPL <= RS, "a pair's rows must fit the row stride");
  static constexpr int kPad = 4;                                 // zero words in front of the planes: "row -1" / "row -2" of the first board
  static constexpr int kState = kPad;                            // [2][kNB5][RS]: black, white
  static constexpr int kMeta = kState + 2 * kNB5 * RS;           // flags[32], last[32], played[32], rng[64]
  static constexpr int kFair = kMeta + 5 * kNB5;                 // [16]: FairShare
  static constexpr int kTmp = kFair + 16;                        // [2][2][RS]: layout change of one pair at load
  static constexpr int kUnion = kTmp + 4 * RS;
  // ply loop: per job its descriptor, per board an info word (what its jobs found); per board the block of the mover's group G and the block in which the
  // opponent groups that are captured or leave M are collected (one OR per job that has such a group); per LANE a seed block
  // that is all zero between two uses (a job's seed is staged through it)
  static constexpr int kZero = kJobCap, kDump = kJobCap + 1;     // slot of an absent job (its class word is zero, never written) / of a write nobody reads
  static constexpr int kCls = kUnion;                            // [kJobCap + 4]: the first kNB5 words are the boards' info words
  static constexpr int kJob = kCls + kJobCap + 4;                // [kJobCap + 4]
  static constexpr int kG = kJob + kJobCap + 4;                  // [kNB5][2][RS]: G, the collected opponent groups
  static constexpr int kSc = kG + 2 * kNB5 * RS;                 // [kWave][RS]
  static constexpr int kLoopEnd = kSc + kWave * RS;
  // load: the v2 analysis in its compact form (region 0 only); tracked boards: the DMA landing area, the parked rows
  static constexpr int kV2 = kUnion;
  static constexpr int kIoEnd = kV2 + (Lds2<R>::kRegion0 > 768 ? Lds2<R>::kRegion0 : 768);
  static constexpr int kDmaWords = (((kNB5 * (5 * R + 1) * 4 + 12 + 15) / 16 + kWave - 1) / kWave) * 256;
  static constexpr int kDmaEnd = kUnion + kDmaWords;
  // byte-plane write-back of a whole group (emit_group)
  static constexpr int kGrpBits = kUnion;
  static constexpr int kGrpWords = ((15 + kNB5 * 6 * R * R + 31) / 32 + 4) & ~3;
  static constexpr int kGrpLut = kGrpBits + kGrpWords;           // uint2[256]
  static constexpr int kGrpEnd = kGrpLut + 512;
  static constexpr int kMax2(int a, int b) { return a > b ? a : b; }
  static constexpr int kTotal = kMax2(kMax2(kLoopEnd, kIoEnd), kMax2(kDmaEnd, kGrpEnd));
  static_assert(kTotal * 4 <= 20480, "two waves per SIMD: 20 KB of LDS per wave");
  static_assert(kUnion % 4 == 0 && kG % 4 == 0 && kSc % 4 == 0, "16-byte alignment of the flood blocks");
  static_assert(3 * kNB5 * RS <= kLoopEnd - kUnion, "parked tracked rows fit the loop area");
};

// set bits of a ballot in the lanes below this one
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
constexpr int QP_L0 = 0xA0;   // quad_perm [0,0,2,2]: the even lane of each pair

// kth_set_bit (gg_v4.h) for TEN rows per lane: the row is found by a search tree over the prefix counts (5 | 2 + 3 | 1 + 1 (+ 1):
// 31 instructions) instead of nine select steps in a row (45); the bit search inside the row is the same.
__device__ __forceinline__ void kth_set_bit10(const uint32_t (&v)[10], const uint32_t (&p)[10], uint32_t tt, int &rr, uint32_t &pos) {
  uint32_t ntt = ~tt;
  // c + ~tt = c - tt - 1 is negative iff tt >= c: the target lies beyond the rows counted by c
  const uint32_t g5 = (uint32_t)((int32_t)(p[4] + ntt) >> 31);
  uint32_t a[5], t[4];
#pragma unroll
  for (int i = 0; i < 5; ++i) a[i] = B3(g5, v[5 + i], v[i], T_SEL);
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = B3(g5, p[5 + i], p[i], T_SEL);
  uint32_t base = g5 & p[4], row = g5 & 5u;
  const uint32_t g2 = (uint32_t)((int32_t)(t[1] + ntt) >> 31);            // rows 2 .. 4 of the half
  const uint32_t b0 = B3(g2, a[2], a[0], T_SEL), b1 = B3(g2, a[3], a[1], T_SEL), u0 = B3(g2, t[2], t[0], T_SEL);
  base = B3(g2, t[1], base, T_SEL);
  row = B3(g2, 2u, row, T_ANDOR);                                          // (0 or 5) + 2: no carry
  const uint32_t g1 = (uint32_t)((int32_t)(u0 + ntt) >> 31);
  uint32_t vr = B3(g1, b1, b0, T_SEL);
  base = B3(g1, u0, base, T_SEL);
  row -= g1;
  const uint32_t g1b = B3(g2, g1, (uint32_t)((int32_t)(t[3] + ntt) >> 31), TA & TB & TC);   // the last row of the three-row group
  vr = B3(g1b, a[4], vr, T_SEL);
  base = B3(g1b, t[3], base, T_SEL);
  row -= g1b;
  ntt += base;                                                          // ~(tt - base)
  uint32_t ps = 0;
#pragma unroll
  for (int sh = 16; sh >= 1; sh >>= 1) {
    const uint32_t e = (uint32_t)__popc((vr >> ps) & ((1u << sh) - 1u)) + ntt;
    const uint32_t ge = (uint32_t)((int32_t)e >> 31);
    ntt = B3(ge, e, ntt, T_SEL);
    ps = B3((uint32_t)sh, ge, ps, T_ANDOR);
  }
  rr = (int)row;
  pos = ps;
}

// flood2_serial (gg_common.h; seeds with their odd rows bit-reversed) for a batch of JOBS of which only some need a fixed
// point: the sweeps go on while a lane with `need` is open.  WEAK: such a lane counts as open only where the fill could still
// grow into a stone that is NOT in `mm` (the rows of M, the stones whose group had >= 2 liberties before the move).  res[] =
// the fill as the last closure test saw it, normal bit order; `open` = what that test found for this lane (0: its fill is
// closed).  A lane whose flood is cut short holds a PART of its group - every liberty of the part is a liberty of the group.
template <int R, bool WEAK>
__device__ __forceinline__ void flood_jobs(const uint32_t (&m)[R], const uint32_t (&mrev)[R], uint32_t (&f)[R], uint32_t (&res)[R],
                                           bool need, const uint32_t (&mm)[R], uint32_t &open) {
  int sweeps = 0;
#pragma unroll 1
  for (int it = 0; it < R * R; ++it) {
#pragma unroll
    for (int r = 0; r < R; ++r) FLOOD_VISIT(r, r - 1, (r & 1) != 0);       // down: domain (r&1) -> ((r+1)&1)
    if (it > 0) {
      uint32_t op = 0, opw = 0, pend = 0, above = 0;
#pragma unroll
      for (int r = R - 1; r >= 0; --r) {
        const uint32_t g = ((r + 1) & 1) ? __brev(f[r]) : f[r];
        res[r] = g;
        if (r < R - 1) {
          const uint32_t t = B3(above, m[r], g, T_AND_ANDN);   // a filled stone below a fillable, unfilled one
          if (WEAK) { op |= t; opw = B3(t, mm[r], opw, (TA & ~TB & 0xFF) | TC); }
          else or_pairs(op, pend, (R - 2 - r) & 1, t);
        }
        above = g;
      }
      if (!WEAK && ((R - 1) & 1)) op |= pend;
      open = op;
      if (__ballot((WEAK ? opw : op) != 0 && need) == 0) { sweeps = 2 * it + 1; break; }
    }
#pragma unroll
    for (int r = R - 1; r >= 0; --r) FLOOD_VISIT(r, r + 1, ((r + 1) & 1) != 0);  // up: domain ((r+1)&1) -> (r&1)
    if (it > 0) {   // (a first test already after the second sweep: 2.23 sweeps per batch, but 27 % of the batches then have a lane to flood on: 1.351 -> 1.385 ms)
      uint32_t op = 0, opw = 0, pend = 0, below = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t g = (r & 1) ? __brev(f[r]) : f[r];
        res[r] = g;
        if (r > 0) {
          const uint32_t t = B3(below, m[r], g, T_AND_ANDN);
          if (WEAK) { op |= t; opw = B3(t, mm[r], opw, (TA & ~TB & 0xFF) | TC); }
          else or_pairs(op, pend, (r - 1) & 1, t);
        }
        below = g;
      }
      if (!WEAK && ((R - 1) & 1)) op |= pend;
      open = op;
      if (__ballot((WEAK ? opw : op) != 0 && need) == 0) { sweeps = 2 * it + 2; break; }
    }
  }
#ifdef GG_AB_SWEEPS
  { int l_; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_));
    if (l_ == 0) { atomicAdd(&gg_sweeps[0], (unsigned long long)sweeps); atomicAdd(&gg_sweeps[1], 1ull); } }
#endif
  (void)sweeps;
}

// liberties (dilate & empty, counted; only min(count, 2) is used) of the group gt[], m[] = the rows of its colour, ot[] = the
// other colour's rows (read from LDS together with m[], BEFORE the flood: read behind it they cost the lane a round trip)
template <int R>
__device__ __forceinline__ uint32_t job_liberties(const uint32_t (&gt)[R], const uint32_t (&ot)[R], const uint32_t (&m)[R]) {
  constexpr uint32_t FULLROW = (1u << R) - 1u;
  uint32_t cnt3[3] = {0u, 0u, 0u};
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t e = B3(ot[r], m[r], FULLROW, ~(TA | TB) & TC & 0xFF);   // empty points
    const uint32_t up = r > 0 ? gt[r - 1] : 0u, dn = r + 1 < R ? gt[r + 1] : 0u;
    const uint32_t dd = B3(shl1(gt[r]), gt[r] >> 1, up, T_OR3);
    const uint32_t l = B3(dd, dn, e, (TA | TB) & TC);
    cnt3[r % 3] += (uint32_t)__popc(l);   // (three accumulating chains, not one of nineteen v_bcnt)
  }
  return cnt3[0] + cnt3[1] + cnt3[2];
}

// job descriptor: bits 0-4 board, 5-13 the seed (flat point index), 15 the colour flooded, 16 the job floods G, 18 the job exists,
// 19-20 the direction of q's neighbour it starts from (0 up, 1 down, 2 left, 3 right)
// info word of a board (cleared in phase 1, ORed by its jobs): bits 0-3 the directions whose opponent group was captured, 4-5 the
// liberties of G (saturated at 2)
template <int R, int IO>
__global__ __launch_bounds__(kWave, 2) void k_rollout5(uint8_t *__restrict__ states, uint64_t *__restrict__ rng,
                                                       int32_t *__restrict__ last_actions, int64_t *__restrict__ steps_done,
                                                       int64_t B, uint32_t inv, int plies, int auto_reset, int nb) {
  static_assert(IO == 0 || IO == 2, "byte planes or tracked boards");
  constexpr int N = R;
  constexpr int RS = Lds5<R>::RS;
  constexpr int RV = (R + 3) / 4;
  constexpr int RPL = Lds5<R>::RPL;
  constexpr int PL = kNB5 * RS;   // words per plane of all boards
  constexpr int ZERO = Lds5<R>::kZero, DUMP = Lds5<R>::kDump;
  constexpr bool TRACKED = IO == 2;
  constexpr int P = N * N, S = 6 * P, W = 5 * N + 1;
  constexpr uint32_t FULLROW = (1u << N) - 1u;
  __shared__ __attribute__((aligned(16))) uint32_t lds[Lds5<R>::kTotal];
  uint32_t *st = lds + Lds5<R>::kState;     // st[colour * PL + board * RS + row]
  uint32_t *flagsv = lds + Lds5<R>::kMeta;  // bit 0 turn, 1 passed, 2 done, 3 on, 5 reset (dirty)
  int *lastv = reinterpret_cast<int *>(lds + Lds5<R>::kMeta + kNB5);
  int *playedv = reinterpret_cast<int *>(lds + Lds5<R>::kMeta + 2 * kNB5);
  uint32_t *rngv = lds + Lds5<R>::kMeta + 3 * kNB5;   // [2 * s], [2 * s + 1]
  uint32_t *tmp = lds + Lds5<R>::kTmp;      // tmp[(half * 2 + set) * RS + row], set 0 = invalid, 1 = M
  uint32_t *clsv = lds + Lds5<R>::kCls;
  uint32_t *jobv = lds + Lds5<R>::kJob;
  uint32_t *gblk = lds + Lds5<R>::kG;
  uint32_t *sc = lds + Lds5<R>::kSc;
  uint32_t *v2 = lds + Lds5<R>::kV2;
  uint32_t *park = lds + Lds5<R>::kUnion;   // tracked I/O: park[set * PL + board * RS + row], set 0 invalid, 1 mb, 2 mw
  const int64_t ngroups = (B + nb - 1) / nb;

  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int64_t b_first = g * nb;
    int ln0;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln0));
    const Half hf = make_half(ln0, N, inv);
    const bool row = hf.hl < RS;
    const int s5 = hf.lane >> 1, t2 = hf.lane & 1, r05 = RPL * t2;   // board / lane of the pair / first row of this lane
    // the next mover's invalid-move mask and the stones of groups with >= 2 liberties, rows r05 .. r05 + RPL - 1 of board
    // s5: in registers from here to the write-back
    uint32_t inv_r[RPL], M[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) inv_r[r] = M[r] = 0u;
    GG_PROF_DECL;
    // ---------------------------------------------------------------- load
    WAVE_SYNC();
    if (TRACKED) {
      // the group's boards are ONE contiguous block of nb x (5 N + 1) words: global -> LDS by LDS-DMA, all of it in flight
      // at once, sorted from the landing area (as k_rollout4 does)
      const int64_t nbrd = (B - b_first) < nb ? (B - b_first) : nb;
      const int nw = (int)nbrd * W;
      const uint32_t *gp = reinterpret_cast<const uint32_t *>(states) + b_first * (int64_t)W;
      constexpr int KD = Lds5<R>::kDmaWords / 256;   // DMA instructions per lane
      const uint8_t *gb = reinterpret_cast<const uint8_t *>(gp);
      const uint32_t mis = (uint32_t)((uintptr_t)gb & 15u);
      const int nvec = (int)((mis + (uint32_t)nw * 4u + 15u) >> 4);
      WAVE_SYNC();
      lds_drain();
      {
        const uint32_t stage_lds = lds_addr(park);
#pragma unroll
        for (int k = 0; k < KD; ++k) {
          const int v = hf.lane + kWave * k;
          if (v < nvec) dma16(gb - mis + 16 * v, stage_lds + 1024u * (uint32_t)k);
        }
      }
      uint64_t xg = 0;
      if (hf.lane < kNB5) xg = rng[(hf.lane < nb && b_first + hf.lane < B) ? b_first + hf.lane : B - 1];
      for (int i = hf.lane; i < 2 * PL; i += kWave) st[i] = 0;      // rows N .. RS-1 and absent boards read as zero
      if (hf.lane < Lds5<R>::kPad) lds[hf.lane] = 0;
      dma_wait();
      WAVE_SYNC();
      const uint32_t *stg = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(park) + mis);   // word i of the block
      for (int i = hf.lane; i < (int)nbrd * 2 * N; i += kWave) {
        const int sb = i / (2 * N), w = i - sb * (2 * N);
        const int pl = w >= N ? 1 : 0;
        st[pl * PL + sb * RS + (w - pl * N)] = stg[sb * W + w];
      }
      if (hf.lane < kNB5) {
        const int sb = hf.lane;
        const bool on = sb < nb && b_first + sb < B;
        flagsv[sb] = on ? ((stg[sb * W + 5 * N] & 7u) | 8u) : 0u;
        lastv[sb] = -1;
        playedv[sb] = 0;
        rngv[2 * sb] = (uint32_t)xg;
        rngv[2 * sb + 1] = (uint32_t)(xg >> 32);
      }
      {
        const bool have = s5 < (int)nbrd;
        const uint32_t *bq = stg + (have ? s5 : 0) * W;
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          const int rw = r05 + r;
          const bool ok = have && rw < N;
          const int rc = ok ? rw : 0;
          const uint32_t iv = bq[2 * N + rc], mb = bq[3 * N + rc], mw = bq[4 * N + rc];
          inv_r[r] = ok ? iv : 0u;
          M[r] = ok ? (mb | mw) : 0u;
        }
      }
      WAVE_SYNC();
    } else {
      if (hf.lane < kNB5) flagsv[hf.lane] = 0;                       // boards beyond nb: off
      for (int i = hf.lane; i < 2 * PL; i += kWave) st[i] = 0;
      if (hf.lane < Lds5<R>::kPad) lds[hf.lane] = 0;
      WAVE_SYNC();
      // Byte planes: pairs of boards, first classes by the per-ply analysis (analyze2); the global loads of pair i + 1 are
      // issued before pair i is converted
      constexpr int NVL = (4 * R * R + 15 + 15) / 512 + 1;   // 16-byte vectors per lane
      static_assert(NVL <= 3, "vectors per lane of a staged board");
      if (hf.lane < kNB5) {   // the generator states of the whole group: one coalesced load
        const uint64_t x = rng[(b_first + hf.lane < B) ? b_first + hf.lane : B - 1];
        rngv[2 * hf.lane] = (uint32_t)x;
        rngv[2 * hf.lane + 1] = (uint32_t)(x >> 32);
      }
      uint4 cv0 = make_uint4(0, 0, 0, 0), cv1 = cv0, cv2 = cv0, nv0 = cv0, nv1 = cv0, nv2 = cv0;
      uint32_t cfb = 0, nfb = 0;
#define GG_ISSUE_PAIR5(I, V0, V1, V2, FB)                                                                              \
      do {                                                                                                             \
        const int s_ = 2 * (I) + hf.h;                                                                                 \
        const int64_t b_ = (b_first + s_ < B) ? b_first + s_ : B - 1;                                                  \
        const uint8_t *gs_ = states + b_ * (int64_t)S;                                                                 \
        FB = 0;                                                                                                        \
        if (hf.hl < 4) {                                                                                               \
          const int off_ = hf.hl == 0 ? 2 * P : hf.hl == 1 ? 3 * P : hf.hl == 2 ? 4 * P : 5 * P;                       \
          FB = gs_[off_];                                                                                              \
        }                                                                                                              \
        const uint32_t mis_ = (uint32_t)((uintptr_t)gs_ & 15u);                                                        \
        const uint4 *ga_ = reinterpret_cast<const uint4 *>(gs_ - mis_);                                                \
        const int nv_ = (int)(mis_ + 4 * P + 15) >> 4;                                                                 \
        if (hf.hl < nv_) V0 = ga_[hf.hl];                                                                              \
        if (NVL > 1 && hf.hl + 32 < nv_) V1 = ga_[hf.hl + 32];                                                         \
        if (NVL > 2 && hf.hl + 64 < nv_) V2 = ga_[hf.hl + 64];                                                         \
      } while (0)
      if (nb >= 2) GG_ISSUE_PAIR5(0, cv0, cv1, cv2, cfb);
#pragma unroll 1
      for (int i = 0; i < nb / 2; ++i) {
        if (i + 1 < nb / 2) GG_ISSUE_PAIR5(i + 1, nv0, nv1, nv2, nfb);
        const int s = 2 * i + hf.h;
        const bool on = b_first + s < B;
        const int64_t b = on ? b_first + s : B - 1;
        uint32_t black, white, invalid, mb = 0, mw = 0;
        const uint8_t *gs = states + b * (int64_t)S;
        uint8_t *io = reinterpret_cast<uint8_t *>(v2) + hf.h * Cfg<R>::kIoBytes;
        const uint32_t mi = (uint32_t)((uintptr_t)gs & 15u);
        const int nv = (int)(mi + 4 * P + 15) >> 4;
        const uint32_t flags = half_of(__ballot(cfb != 0), hf.h) & 0xFu;   // bit 0 turn, 1 (unused), 2 passed, 3 done
        WAVE_SYNC();
        uint4 *iov = reinterpret_cast<uint4 *>(io);
        if (hf.hl < nv) iov[hf.hl] = cv0;
        if (NVL > 1 && hf.hl + 32 < nv) iov[hf.hl + 32] = cv1;
        if (NVL > 2 && hf.hl + 64 < nv) iov[hf.hl + 64] = cv2;
        WAVE_SYNC();
        black = plane_to_row<R>(io + mi, N, hf.hl);
        white = plane_to_row<R>(io + mi + P, N, hf.hl);
        invalid = plane_to_row<R>(io + mi + 3 * P, N, hf.hl);
        const uint32_t turn = flags & 1u, passed = (flags >> 2) & 1u, done = (flags >> 3) & 1u;
        uint32_t ab;
        analyze2<R, false>(black, white, hf.full_l1 & ~(black | white), hf, v2, mb, ab, mw, nullptr, nullptr, true);
        if (row) {
          st[0 * PL + s * RS + hf.hl] = black;
          st[1 * PL + s * RS + hf.hl] = white;
          tmp[(hf.h * 2 + 0) * RS + hf.hl] = invalid;
          tmp[(hf.h * 2 + 1) * RS + hf.hl] = mb | mw;
        }
        if (hf.hl == 0) {
          flagsv[s] = turn | (passed << 1) | (done << 2) | (on ? 8u : 0u);
          lastv[s] = -1;
          playedv[s] = 0;
        }
        WAVE_SYNC();
        if ((hf.lane >> 2) == i) {   // the two pairs of lanes that own these boards pick their rows up
          const uint32_t *tp = tmp + ((s5 & 1) * 2) * RS + r05;
#pragma unroll
          for (int r = 0; r < RPL; ++r) {
            inv_r[r] = tp[r];
            M[r] = tp[RS + r];
          }
        }
        WAVE_SYNC();
        cv0 = nv0; cv1 = nv1; cv2 = nv2; cfb = nfb;
      }
#undef GG_ISSUE_PAIR5
    }
    // the slot of an absent job: an all-zero block and class word (the loop area was the load's scratch)
    WAVE_SYNC();
    for (int i = hf.lane; i < kWave * RS; i += kWave) sc[i] = 0u;   // the seed blocks: all zero between two uses
    WAVE_SYNC();

    // the flag word, the generator and the played plies of this lane's board travel in REGISTERS through the plies (the same
    // in both lanes of a pair): read back from LDS every ply they cost phase 1 a dependent round trip; LDS keeps the copies
    // the write-back reads (stores only)
    uint32_t flr = flagsv[s5];
    const uint64_t x0r = ((uint64_t)rngv[2 * s5 + 1] << 32) | rngv[2 * s5];
    int playedr = 0;
    // ---------------------------------------------------------------- the plies
    GG_PROF(6);   // load
    FairShare fair(lds + Lds5<R>::kFair);
    const uint32_t fair_lag = plies >= 192 ? 24u : (plies >= 16 ? (uint32_t)plies >> 3 : 2u);
    bool lead = false;   // this wave is >= fair_lag plies ahead of its SIMD-mate
    uint32_t uq = 0;   // this lane's pre-mixed draw: lane j of a pair holds the one of ply (t & ~1) + j, swapped every ply
#pragma unroll 1
    for (int t = 0; t < plies; ++t) {
      // Fair share of the SIMD (gg_common.h) every fourth ply - without it the older of a SIMD's two waves runs ahead and the
      // launch ends on one wave per SIMD: 1.51 -> 1.66 ms - combined with the PHASE of the ply: the flood of phase 2b is one
      // dependent chain per lane that needs the issue port every fifth cycle or so, the other phases have ten independent
      // rows per lane.  A wave in the flood therefore yields (priority 0 / 1: leader / straggler) and a wave in any other phase
      // issues first (2 / 3): 1.509 -> 1.488 ms per launch of 65 536 games x 256 plies; the other way round 1.561.
      if ((t & 3) == 0 && plies >= 8) {
        const uint32_t left = (uint32_t)(plies - t);
        lead = fair.behind((uint32_t)t, left < fair_lag ? (left > 2u ? left : 2u) : fair_lag) != 0u;
        if (lead) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
      }
      int ln;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      const int s4 = (ln >> 1) & 31, t5 = ln & 1, r0 = RPL * t5;   // board, lane of the pair, first row of this lane
      const bool bl = s4 < nb;
      uint32_t full[RPL];   // the N-bit row mask of the lane's rows that exist
#pragma unroll
      for (int r = 0; r < RPL; ++r) full[r] = (r0 + r < N) ? FULLROW : 0u;

      int a_q;
      uint32_t fl_q;
      // phase 1 - two lanes per board, RPL rows each: liveness, the draw, the k-th valid point of the mask
      {
        const uint32_t fl = flr;
        const bool on = bl && ((fl >> 3) & 1u);
        const bool done = (fl >> 2) & 1u;
        const bool live = on && !(done && !auto_reset);
        const bool reset = live && done;           // auto-reset: the board is init_state from now on
        uint32_t v[RPL], p[RPL];
        const uint32_t rm = reset ? ~0u : 0u;   // a board being reset plays on the empty board
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          v[r] = B3(full[r], rm, inv_r[r], TA & (TB | (~TC & 0xFF)));
          p[r] = (uint32_t)__popc(v[r]) + (r ? p[r - 1] : 0u);
        }
        const uint32_t T = p[RPL - 1];
        // valid points of the board before this lane's rows (Pb) and on the whole board (n): one swap inside the pair
        const uint32_t oth = dpp0<QP_X1>(T);
        const uint32_t Pb = t5 ? oth : 0u, n = T + oth;
        // The draws of a board, TWO plies at a time: the draw of ply t is mix(x0 + (t + 1) c) with x0 the generator the
        // launch found; lane j of the pair mixes the draw of ply t + j every second ply, a ply takes the even lane's and
        // the pair swaps
        if ((t & 1) == 0) {
          uint64_t xx = x0r + (uint64_t)(uint32_t)(t + t5) * 0x9E3779B97F4A7C15ull;
          uq = (uint32_t)(splitmix_next(xx) >> 32);
        }
        const uint32_t uh = dpp0<QP_L0>(uq);
        uq = dpp0<QP_X1>(uq);
        const uint32_t k = __umulhi(uh, n + 1u);   // k == n: the pass
        const bool hit = k >= Pb && k < Pb + T;       // this lane holds the k-th valid point
        int rr;
        uint32_t pos;
        if constexpr (RPL == 10) kth_set_bit10(v, p, (k - Pb) & 0x3FFu, rr, pos);
        else kth_set_bit<RPL>(v, p, (k - Pb) & 0x3FFu, rr, pos);
        const int rabs = r0 + rr;
        // (k < n: exactly one lane of the pair holds the point and hands it to the other; k, n and live are the same in both)
        const uint32_t cand = hit ? (uint32_t)(rabs * N + (int)pos) : 0u;
        const uint32_t pt = cand | dpp0<QP_X1>(cand);
        a_q = !live ? -1 : (k < n ? (int)pt : P);
        const bool place = live && hit;
        fl_q = reset ? 40u : fl;   // a board being reset: on, dirty, black to move
        uint64_t resetm = __ballot(reset && t5 == 0);
        const bool none_live = __ballot(live) == 0;
        if (none_live && resetm == 0) break;
        if (resetm) {   // rare
          if (reset) {
#pragma unroll
            for (int r = 0; r < RPL; ++r) inv_r[r] = M[r] = 0u;
          }
          while (resetm) {
            const int s = (__ffsll((unsigned long long)resetm) - 1) >> 1;   // lane 2 s -> board s
            resetm &= resetm - 1;
            for (int i = hf.lane; i < 2 * RS; i += kWave) st[(i / RS) * PL + s * RS + (i % RS)] = 0;
            if (hf.lane == 0) flagsv[s] = 8u | 32u;   // on, reset (written back even if nothing is played)
          }
          if (none_live) break;
        }
        WAVE_SYNC();
        // the new stone goes into the mover's plane right away, and - as the group G it forms on its own - into the board's
        // G block (a job floods over it when q has a friendly neighbour)
        // (the lanes of EVERY board clear its G block and its collection block - a board that passes or idles leaves them
        // empty, phase 3 reads them unmasked -, then the lane that holds the point writes the stone: DS instructions of a wave
        // execute in order)
        uint32_t *gb = gblk + 2 * s4 * RS;
        if (t5 == 0) clsv[s4] = 0u;   // the board's info word
        {
          uint4 *pz = reinterpret_cast<uint4 *>(gb + t5 * RS);   // (lane 0 of the pair: the G block, lane 1: the collection block)
#pragma unroll
          for (int i = 0; i < RV; ++i) pz[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        asm volatile("" ::: "memory");
        if (place) {
          const int turn = reset ? 0 : (int)(fl & 1u);
          atomicOr(st + turn * PL + s4 * RS + rabs, 1u << pos);   // (ds_or without a return value: no round trip inside the phase)
          gb[rabs] = 1u << pos;
        }
      }
      WAVE_SYNC();
      GG_PROF(0);

      // phase 2a - the board's lanes look at q's neighbours: lane 0 of the pair at the ones above / below, lane 1 at the ones
      // to the left / right; what the board-level tests need (empty neighbours of q, is any of them friendly, is q boxed
      // in) is one packed pair sum; the floods become jobs
      uint32_t qs;                 // bits 0-2 empty neighbours of q, 11 q has a friendly neighbour, 19 q is NOT boxed in
      int njobs;
      {
        const int a = a_q;
        const uint32_t turn = fl_q & 1u;
        const uint32_t mv1 = ((uint32_t)a < (uint32_t)P) ? 1u : 0u;   // a stone was placed
        int ar, ac;
        split_action(mv1 ? a : 0, N, inv, ar, ac);
        const uint32_t *pm = st + turn * PL + s4 * RS, *po = st + (1u - turn) * PL + s4 * RS;
        uint32_t obit[2], packed = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int sg = 2 * j - 1;
          const int dr = t5 ? 0 : sg, dc = t5 ? sg : 0;
          const int nr = ar + dr, nc = ac + dc;   // row -1 .. N, column -1 .. N
          // (row -1 of a board is a zero row of the board before it or the pad, row N a zero row)
          const uint32_t rowm = pm[nr], rowo = po[nr];
          const uint32_t ncs = (uint32_t)nc & 31u;   // column -1 reads bit 31, column N bit N: never set in a row
          const uint32_t mbit = (rowm >> ncs) & mv1, ob = (rowo >> ncs) & mv1;
          const uint32_t onb = ((uint32_t)nr < (uint32_t)N && (uint32_t)nc < (uint32_t)N) ? mv1 : 0u;
          const uint32_t ex = mbit | ob;
          packed += (onb & ~ex) | (mbit << 8) | ((onb & ~ob) << 16);
          obit[j] = ob;
        }
        qs = packed + dpp0<QP_X1>(packed) + 0x70700u;
        const uint32_t friendly = (qs >> 11) & 1u;
        const uint32_t gf = t5 ? 0u : friendly;             // the pair's even lane posts the G job
        const uint32_t c = gf + obit[0] + obit[1];
        const uint64_t b0 = __ballot((c & 1u) != 0u), b1 = __ballot((c & 2u) != 0u);
        const uint32_t base = mbcnt64(b0) + 2u * mbcnt64(b1);
        njobs = (int)__popcll(b0) + 2 * (int)__popcll(b1);
        const uint32_t sG = base, s0 = base + gf, s1 = s0 + obit[0];
        // (the seed travels as a flat point index: q itself for G, q -+ N / q -+ 1 for the opponent stone above / below / left / right)
        const uint32_t common = (uint32_t)s4 | (1u << 18) | ((turn ^ 1u) << 15);
        const int step = t5 ? 1 : N;
        jobv[gf ? sG : (uint32_t)DUMP] = ((uint32_t)s4 | (1u << 18) | (turn << 15) | (1u << 16)) | ((uint32_t)a << 5);
        const uint32_t dirs = (uint32_t)t5 << 20;   // bits 19-20: the direction of the job (0 up, 1 down, 2 left, 3 right)
        jobv[obit[0] ? s0 : (uint32_t)DUMP] = common | dirs | ((uint32_t)(a - step) << 5);
        jobv[obit[1] ? s1 : (uint32_t)DUMP] = common | dirs | (1u << 19) | ((uint32_t)(a + step) << 5);
      }
      WAVE_SYNC();

      // phase 2b - lane L runs job L: the flood (seed staged through the job's cleared block), then the liberties of the
      // group (dilate & empty, saturated at 2), all rows in registers; the class word: bits 0-1 liberties, 5 an opponent group
      // without a liberty (captured).  An opponent group that keeps >= 2 liberties zeroes
      // its block: phase 3 never sees it.
      if (plies >= 8) { if (lead) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }
#pragma unroll 1
      for (int jb = 0; jb < njobs; jb += kWave) {
        const int j = jb + ln;
        const bool have = j < njobs;
        const uint32_t d = jobv[have ? j : DUMP];
        const uint32_t ex = have ? 1u : 0u;
        const int sj = (int)(d & 31u);
        int sr, scol;
        split_action((int)((d >> 5) & 511u), N, inv, sr, scol);
        const uint32_t ownc = (d >> 15) & 1u;
        const uint32_t isG = have ? (d >> 16) & 1u : 0u;
        uint32_t *blk = sc + ln * RS;   // this lane's seed block (all zero)
        const uint4 *lds4 = reinterpret_cast<const uint4 *>(lds);
        const uint4 *pmv = lds4 + (Lds5<R>::kState + (int)ownc * PL + sj * RS) / 4;
        const uint4 *pov = lds4 + (Lds5<R>::kState + (int)(ownc ^ 1u) * PL + sj * RS) / 4;
        // the rows of M (the classes BEFORE this move) of the job's board, out of the registers of the two lanes that hold them
        uint32_t mm[R];
        {
          const int src = 8 * sj;   // byte address of lane 2 sj for ds_bpermute
#pragma unroll
          for (int i = 0; i < RPL; ++i) {
            mm[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)M[i]);
            if (RPL + i < R) mm[RPL + i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src + 4, (int)M[i]);
          }
        }
        uint32_t cnt = 0;
        uint32_t res[R];   // the group (or the part of it that settles its class), normal bit order
        {
          uint32_t m[R], mrev[R], f[R], ot[R];
          {
            uint32_t mt[RV * 4], ft[RV * 4];
#pragma unroll
            for (int i = 0; i < RV; ++i) {
              const uint4 x = pmv[i], y = pov[i];
              mt[4 * i] = x.x; mt[4 * i + 1] = x.y; mt[4 * i + 2] = x.z; mt[4 * i + 3] = x.w;
              ot[4 * i] = y.x;
              if (4 * i + 1 < R) ot[4 * i + 1] = y.y;
              if (4 * i + 2 < R) ot[4 * i + 2] = y.z;
              if (4 * i + 3 < R) ot[4 * i + 3] = y.w;
            }
            // the seed is one bit of one row: written into the lane's zero block at its (dynamic) row and read back as the
            // flood's row set - two LDS instructions instead of a select per row (odd rows bit-reversed) -, then cleared again
            {
              const int srw = sr & (int)(0u - ex);
              asm volatile("" ::: "memory");
              blk[srw] = ex << (((uint32_t)scol ^ (0u - ((uint32_t)sr & 1u))) & 31u);   // odd rows: bit 31 - scol
              asm volatile("" ::: "memory");
              const uint4 *pf = reinterpret_cast<const uint4 *>(blk);
#pragma unroll
              for (int i = 0; i < RV; ++i) {
                const uint4 x = pf[i];
                ft[4 * i] = x.x; ft[4 * i + 1] = x.y; ft[4 * i + 2] = x.z; ft[4 * i + 3] = x.w;
              }
              asm volatile("" ::: "memory");
              blk[srw] = 0u;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
              m[r] = mt[r];
              mrev[r] = __brev(m[r]);
              f[r] = ft[r];
              res[r] = 0u;
            }
          }
          GG_PROF(1);
          // Which floods must reach their fixed point inside the batch's loop?  (Its length is the longest of its floods: 4.21
          // sweeps per batch when all 1.56 floods per board count, 3.87 when only G's 0.6 per board do, 3.00 - the minimum - with
          // the weak closure below, tools/exp/r5_sweeps.py: 1.445 -> 1.415 -> 1.349 ms per launch of 65 536 games x 256 plies;
          // 1.8 % of the batches have a lane that floods on afterwards.)
          //  * An OPPONENT group never: cut short, the part found so far either has two liberties - liberties of the whole
          //    group, which keeps its class: phase 3 never sees it - or fewer, and then the group may be captured or leave M
          //    and its full extent matters: that lane floods on afterwards (below; groups with < 2 liberties are small).
          //  * The mover's group G only as far as its stones OUTSIDE M go: with two liberties found G joins M whole, and what the
          //    cut-short flood has not reached of it are stones of groups that were in M already (a group in atari that q
          //    connects hangs on q itself, stone by stone outside M: the weak closure holds it whole); with fewer, as above.
          // (the two-chain flood2_dual: 1.758 against 1.579 ms per launch - one more sweep-equivalent, as in k_rollout4)
          uint32_t open = 0;
          flood_jobs<R, true>(m, mrev, f, res, isG != 0u, mm, open);
          GG_PROF(2);
          cnt = job_liberties<R>(res, ot, m);
          const bool unsettled = have && open != 0u && cnt < 2u;
          if (__ballot(unsettled)) {
            // (the sweeps resume from the fill as the last test left it: normal bit order -> odd rows reversed)
#pragma unroll
            for (int r = 0; r < R; ++r) f[r] = (r & 1) ? __brev(res[r]) : res[r];
            flood_jobs<R, false>(m, mrev, f, res, unsettled, mm, open);
            cnt = job_liberties<R>(res, ot, m);
          }
        }
        const uint32_t lib2 = cnt < 2u ? cnt : 2u;
        // G goes to its board's G block (over the stone phase 1 left there); an opponent group with no liberty left
        // (captured) or with one (it leaves M) is ORed into the board's collection block - one that keeps >= 2 is dropped
        uint32_t *gb = gblk + 2 * sj * RS;
        if (isG) {
          uint4 *pz = reinterpret_cast<uint4 *>(gb);
#pragma unroll
          for (int i = 0; i < RV; ++i)
            pz[i] = make_uint4(res[4 * i], 4 * i + 1 < R ? res[4 * i + 1] : 0u, 4 * i + 2 < R ? res[4 * i + 2] : 0u, 4 * i + 3 < R ? res[4 * i + 3] : 0u);
          if (lib2) atomicOr(clsv + sj, lib2 << 4);
        } else if (have && cnt < 2u) {
#pragma unroll
          for (int r = 0; r < R; ++r) atomicOr(gb + RS + r, res[r]);
          if (cnt == 0u) atomicOr(clsv + sj, 1u << ((d >> 19) & 3u));
        }
      }
      if (plies >= 8) { if (lead) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3); }
      WAVE_SYNC();
      GG_PROF(3);

      // phase 3 - all thirty-two boards in ONE pass, RPL adjacent rows per lane: patch the classes, resolve captures and
      // ko, the next mover's mask (k_rollout4's phase 3 on pairs; A = this lane's two directions, B = its partner's)
      {
        const int a = a_q;
        const uint32_t fl = fl_q;
        const uint32_t info = clsv[s4];
        const int turn0 = fl & 1u;
        uint32_t *pmine = st + turn0 * PL + s4 * RS + r0;
        uint32_t *popp = st + (1 - turn0) * PL + s4 * RS + r0;
        const uint32_t *gG = gblk + 2 * s4 * RS + r0;   // the board's G block, behind it the collected opponent groups
        uint32_t mine1[RPL], opp0[RPL], all4[RPL], bg[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          mine1[r] = pmine[r];   // (rows >= N are zero)
          opp0[r] = popp[r];
          bg[r] = gG[r];
          all4[r] = gG[RS + r];
        }
        const bool moves_now = a >= 0;
        const bool is_pass = a == P;
        int ar, ac;
        split_action(a, N, inv, ar, ac);                       // (garbage for a pass / an idle board: masked below)
        const uint32_t capt_m = info & 15u;   // the directions in which an opponent group died (never set on a board that does not move)
        // The collection block holds the opponent groups next to q with NO liberty left (captured: q was their only liberty, so
        // they were never in M) or with exactly ONE (they had q and one more: they were in M).  So it splits by M alone:
        // captured = all4 & ~M, leaving M = all4 & M.
        uint32_t g0[RPL], cap[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          g0[r] = bg[r];   // the G block: the flood of G, the stone alone as phase 1 left it there, or nothing (pass / idle board)
          cap[r] = B3(all4[r], M[r], M[r], TA & ~TB & 0xFF);
        }
        // liberties of G among the empty points (saturated at 2): G's own count, or the empty neighbours of q when the
        // stone stands alone
        const uint32_t ne = qs & 7u, ne2 = ne < 2u ? ne : 2u;
        uint32_t libsG = ((qs >> 11) & 1u) ? ((info >> 4) & 3u) : ne2;
        uint32_t ko_oh = 0, ko_bit = 0;   // the ko point: one-hot row of this lane / column bit (almost always none)
        if (__ballot(capt_m != 0u)) {   // a capture on some board of the wave
          const uint32_t ncapn = (uint32_t)__popc(capt_m);        // captured neighbours of q
          if (__ballot(ncapn == 1u && libsG == 0u)) {
            uint32_t dg[RPL];
            dilate_rows<RPL>(g0, dg);
            uint32_t cntc = 0;
#pragma unroll
            for (int r = 0; r < RPL; ++r) cntc += (uint32_t)__popc(dg[r] & cap[r]);
            const uint32_t c2 = cntc < 2u ? cntc : 2u;
            const uint32_t tot = c2 + dpp0<QP_X1>(c2);
            libsG += (ncapn == 1u && libsG == 0u) ? (tot < 2u ? tot : 2u) : ncapn;
          } else {
            libsG += ncapn;
          }
          // gogame.py:72-75: ko iff exactly one stone died and the new stone is boxed in (one captured NEIGHBOUR and a boxed-in
          // stone first: rare enough to keep the rest off the usual path)
          const bool ko1 = ncapn == 1u && !(qs & CL_OPEN);
          if (__ballot(ko1)) {
            uint32_t died = 0;   // captured stones on this lane's rows (one captured neighbour: exactly one stone died iff its group is that stone)
#pragma unroll
            for (int r = 0; r < RPL; ++r) died += (uint32_t)__popc(cap[r]);
            const bool ko = ko1 && died + dpp0<QP_X1>(died) == 1u;
            // the one captured stone is q's neighbour in the direction of its job (bit 0 up, 1 down, 2 left, 3 right)
            const uint32_t kr = (uint32_t)ar - (capt_m & 1u) + ((capt_m >> 1) & 1u) - (uint32_t)r0;
            ko_oh = (ko && kr < (uint32_t)RPL) ? (1u << (kr & 31)) : 0u;
            ko_bit = 1u << (((uint32_t)ac - ((capt_m >> 2) & 1u) + (capt_m >> 3)) & 31u);
          }
          // the mover's groups in atari next to a captured stone (and not merged into G) now have >= 2 liberties: they join M
          // before the classes are patched
          uint32_t atari[RPL];
          uint32_t anya = 0;
#pragma unroll
          for (int r = 0; r < RPL; ++r) { atari[r] = B3(mine1[r], M[r], g0[r], TA & ~(TB | TC) & 0xFF); anya |= atari[r]; }
          if (__ballot(anya != 0u && capt_m != 0u)) {
            uint32_t f[RPL], anyf = 0;
            dilate_rows<RPL>(cap, f);
#pragma unroll
            for (int r = 0; r < RPL; ++r) { f[r] &= atari[r]; anyf |= f[r]; }
            if (__ballot(anyf != 0)) {
#pragma unroll 1
              for (int it = 0; it < R * R; ++it) {
                uint32_t dd[RPL], chg = 0;
                dilate_rows<RPL>(f, dd);
#pragma unroll
                for (int r = 0; r < RPL; ++r) {
                  const uint32_t nw = B3(dd[r], atari[r], f[r], T_ANDOR);
                  chg |= nw ^ f[r];
                  f[r] = nw;
                }
                if (__ballot(chg != 0) == 0) break;
              }
#pragma unroll
              for (int r = 0; r < RPL; ++r) M[r] |= f[r];
            }
          }
        }
        const uint32_t gsel = libsG >= 2u ? ~0u : 0u;
        uint32_t Mo2[RPL], opp1[RPL], Mm2[RPL], e[RPL], x[RPL], nbr[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          Mo2[r] = B3(M[r], opp0[r], all4[r], TA & TB & ~TC & 0xFF);        // (a captured group was never in M)
          opp1[r] = B3(opp0[r], cap[r], cap[r], TA & ~TB & 0xFF);
          const uint32_t Mm = B3(M[r], mine1[r], g0[r], TA & TB & ~TC & 0xFF);
          Mm2[r] = B3(gsel, g0[r], Mm, T_ANDOR);                           // (M & mine & ~g0) | (gsel & g0)
          e[r] = B3(full[r], opp1[r], mine1[r], TA & ~(TB | TC) & 0xFF);
          x[r] = B3(mine1[r], Mm2[r], e[r], (TA & ~TB & 0xFF) | TC) | Mo2[r];
        }
        dilate_rows<RPL>(x, nbr);
        const uint32_t mv_m = moves_now ? ~0u : 0u;
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          const uint32_t invalid = B3(e[r], nbr[r], full[r], ~(TA & TB) & TC & 0xFF);
          inv_r[r] = B3(mv_m, invalid, inv_r[r], T_SEL);
          M[r] = Mm2[r] | Mo2[r];
        }
        if (__ballot(ko_oh != 0u)) {   // (only a board that moved has a ko point)
#pragma unroll
          for (int r = 0; r < RPL; ++r) inv_r[r] |= (uint32_t)__builtin_amdgcn_sbfe((int)ko_oh, r, 1) & ko_bit;
        }
        if (capt_m) {
#pragma unroll
          for (int r = 0; r < RPL; ++r) popp[r] = opp1[r];
        }
        // (a board being reset found fl_q = on | dirty | black to move in phase 1: the register copy follows even if it does not move)
        flr = fl;
        if (moves_now) {
          const uint32_t passed0 = (fl >> 1) & 1u, done0 = (fl >> 2) & 1u;
          const uint32_t passed = is_pass ? 1u : 0u, done = done0 | (passed & passed0);
          flr = (uint32_t)(turn0 ^ 1) | (passed << 1) | (done << 2) | 8u | (fl & 32u);
          playedr += 1;
          if (t5 == 0) {
            flagsv[s4] = flr;
            lastv[s4] = a;
            playedv[s4] = playedr;
          }
        }
      }
      WAVE_SYNC();
      GG_PROF(4);
    }
    if (plies >= 8) fair.release();
    GG_PROF(5);

    // ---------------------------------------------------------------- store
    int lnS;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lnS));
    const int s5s = lnS >> 1, r05s = RPL * (lnS & 1);
    WAVE_SYNC();
    if (TRACKED) {
      // park the register rows, then one flat coalesced copy of the group's contiguous block
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        if (r05s + r < RS) {
          const uint32_t bk = st[0 * PL + s5s * RS + r05s + r], wh = st[1 * PL + s5s * RS + r05s + r];
          park[0 * PL + s5s * RS + r05s + r] = inv_r[r];
          park[1 * PL + s5s * RS + r05s + r] = M[r] & bk;
          park[2 * PL + s5s * RS + r05s + r] = M[r] & wh;
        }
      }
      WAVE_SYNC();
      const int64_t nbrd = (B - b_first) < nb ? (B - b_first) : nb;
      const int nw = (int)nbrd * W;
      uint32_t *gp = reinterpret_cast<uint32_t *>(states) + b_first * (int64_t)W;
      // (untouched boards are not rewritten: one bit per board, read once; four words per lane and round with their LDS reads in
      // flight together - as k_rollout4's write-back: 48 rounds of flag read -> branch -> row read -> store were 10 us of a launch,
      // now 7.5)
      bool tch = false;
      if (lnS < (int)nbrd) tch = playedv[lnS] != 0 || (flagsv[lnS] & 32u);
      const uint64_t tmask = __ballot(tch);
#pragma unroll 1
      for (int i0 = lnS; i0 < nw; i0 += 4 * kWave) {
        uint32_t v[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k * kWave;
          const int ic = i < nw ? i : 0;
          const int sb = ic / W, w = ic - sb * W;
          ok[k] = i < nw && ((tmask >> sb) & 1ull);
          const int pl = w / N, rw = w - pl * N;   // (w == 5 N: pl == 5, rw == 0)
          const uint32_t *src = pl >= 5 ? flagsv + sb : (pl < 2 ? st + pl * PL + sb * RS + rw : park + (pl - 2) * PL + sb * RS + rw);
          const uint32_t x = *src;
          v[k] = pl >= 5 ? (x & 7u) : x;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (ok[k]) gp[i0 + k * kWave] = v[k];
      }
      if (lnS < nb && b_first + lnS < B) {
        const int sb = lnS;
        const int64_t b = b_first + sb;
        const int played = playedv[sb];
        rng[b] = (((uint64_t)rngv[2 * sb + 1] << 32) | rngv[2 * sb]) + (uint64_t)(uint32_t)played * 0x9E3779B97F4A7C15ull;
        if (last_actions) last_actions[b] = lastv[sb];
        if (steps_done && played) atomicAdd(reinterpret_cast<unsigned long long *>(steps_done) + b, (unsigned long long)played);
      }
      WAVE_SYNC();
    } else {
      // byte planes in place: the whole group in one contiguous write, then the per-game outputs
      const int nbrd = (int)((B - b_first) < nb ? (B - b_first) : nb);
      bool any_wr = false;
      if (lnS < nbrd) any_wr = playedv[lnS] != 0 || (flagsv[lnS] & 32u);
      // (the per-game words are read before the emitter takes the loop area over: the meta words live outside it)
      if (__ballot(any_wr))
        emit_group<R, RPL, 2>(states + b_first * (int64_t)S, nbrd, N, st, PL, RS, inv_r, flagsv, lds + Lds5<R>::kGrpBits,
                              reinterpret_cast<uint2 *>(lds + Lds5<R>::kGrpLut), lnS);
      if (lnS < nbrd) {
        const int sb = lnS;
        const int64_t b = b_first + sb;
        const int played = playedv[sb];
        rng[b] = (((uint64_t)rngv[2 * sb + 1] << 32) | rngv[2 * sb]) + (uint64_t)(uint32_t)played * 0x9E3779B97F4A7C15ull;
        if (last_actions) last_actions[b] = lastv[sb];
        if (steps_done && played) atomicAdd(reinterpret_cast<unsigned long long *>(steps_done) + b, (unsigned long long)played);
      }
      WAVE_SYNC();
    }
    GG_PROF(7);   // write-back
    GG_PROF_FLUSH;
  }
}

}  // namespace gg
