// gg_v4.h - the multi-ply kernel: SIXTEEN BOARDS PER WAVEFRONT (one DPP quad of lanes per board), liberty classes
// carried from ply to ply in registers.
#pragma once
#include "gg_v2.h"

namespace gg {

// ===================================================================== v4: incremental analysis, 16 boards per wave
// A per-ply kernel (gg_v2.h) re-derives every group's liberty class each ply with 22 floods per board.  The cost of a
// flood batch does not depend on how many of the wave's 64 lanes carry a flood, so the lever is floods per board: a
// move at q only changes the groups ADJACENT to q (and, rarely, to a captured group).  The kernel keeps, per board,
// the stones whose group has >= 2 liberties (`M`; every other stone is in atari: a legal position has no
// liberty-less group) and the next mover's invalid-move mask, and updates them per ply from at most FOUR floods:
//   * the mover's group G that the new stone joins - only when q has a friendly neighbour (otherwise G = {q}, whose
//     liberties are q's empty neighbours: no flood);
//   * the opponent's group at each neighbour of q that holds an opponent stone.
// Four lanes always suffice: lane t of a board's quad takes the neighbour of q in direction t (up, down, left, right) -
// an opponent stone: the lane floods its group; a friendly stone: the lane floods G from q into the board's G block
// (every friendly lane the same flood, a benign duplicate); nothing: an empty flood.  (Round 3 compacted the opponent
// neighbours onto lanes 0.. and kept G on lane 3, which cost more instructions than two floods.)  Round 1's kernel gave every
// board five fixed roles (12 boards x 5 lanes); with four lanes a board is exactly one DPP QUAD - every
// board-level reduction is two quad_perm moves - 16 boards share a flood batch instead of 12, and 65 536 games are
// exactly 4 096 waves = ONE resident set of an MI355X (256 CUs x 4 SIMDs x 4 waves): no second, part-filled round.
// A ply is three phases on ONE lane assignment (board = lane / 4, t = lane % 4; in phases 1 and 3 lane t owns the
// RPL = ceil(R / 4) adjacent rows RPL t .. RPL t + RPL - 1):
//   1. sampling: liveness, the generator (drawn redundantly by the four lanes), the k-th valid point of the mask (a
//      lane counts its own rows, the board's prefix / total come from a two-step quad scan, the lane that holds the
//      point selects row and bit), the stone ORed into the mover's plane; auto-reset on a rare path;
//   2. one lane per (board, direction): what its neighbour of q holds (two row reads), the flood (seed staged through
//      the lane's cleared flood block), then the liberties (dilate & empty, saturated at 2) of the lane's group, all
//      rows in registers; the class word carries them together with what the quad learnt about q (one packed quad sum:
//      empty neighbours, any friendly one, boxed in); an opponent group that keeps >= 2 liberties zeroes its result,
//      so phase 3 never sees it;
//   3. class patch, all sixteen boards in one pass: an opponent group next to q with no liberty left is captured, with
//      one left it leaves M; G takes the class of its own count (+ the captured points next to it); a mover's group in
//      atari next to a captured stone gains a liberty (rare; a flood through the atari set in this layout); every
//      other group keeps its class.  The invalid-move mask follows from the classes exactly as in the per-ply kernels.
// M and the mask are produced and consumed by the same lanes (phases 3 -> 1 -> 3), so they live in REGISTERS for the
// whole launch (2 x RPL VGPRs); LDS holds the two stone planes (which the flood lanes read in the other layout), the
// flood results (five blocks per board: one per lane + G) and a few words per board: 10 000 B per wave at 19x19 (the limit
// for four waves per SIMD is 10 240 B).  The action and the flag word of a ply travel from phase 1 to phases 2 and 3 in
// registers; tracked boards enter a launch by LDS-DMA.
constexpr int kNB4 = 16;

// gg_ws.h (policy-weighted sampling, one DPP row of 16 lanes per board); used by the weighted env step below
template <int NJ> __device__ __forceinline__ int wsample_row(const uint32_t (&bits)[NJ], const uint32_t (&vm)[NJ], uint32_t uhi, int lane);
template <int NJ, int AMIN> __device__ __forceinline__ void wload_row(const void *__restrict__ wbase, int64_t board, int wt, int A, int lane, uint32_t (&bits)[NJ]);
template <int NJ> __device__ __forceinline__ void wmask_from_bits(const uint32_t *vw, int lane, uint32_t (&vm)[NJ]);

#ifdef GG_AB_WHERE
// A/B builds only: per workgroup (XCC, HW_ID, duration in 100 MHz ticks) of k_rollout4 (the array lives in gg_v2.h)
#define GG_WHERE_BEGIN const long long tw0_ = wall_clock64()
#define GG_WHERE_END do { if (threadIdx.x == 0 && blockIdx.x < 16384) { unsigned int hw_, xc_; \
    asm volatile("s_getreg_b32 %0, hwreg(4)" : "=s"(hw_)); asm volatile("s_getreg_b32 %0, hwreg(20)" : "=s"(xc_)); \
    gg_where[3 * blockIdx.x] = xc_; gg_where[3 * blockIdx.x + 1] = hw_; gg_where[3 * blockIdx.x + 2] = (unsigned int)(wall_clock64() - tw0_); } } while (0)
#else
#define GG_WHERE_BEGIN do {} while (0)
#define GG_WHERE_END do {} while (0)
#endif

template <int R>
struct Lds4 {
  static constexpr int RS = Cfg<R>::kRowStride;
  static constexpr int RPL = (R + 3) / 4;                        // rows per lane in phases 1 and 3
  static_assert(4 * RPL <= RS, "a quad's rows must fit the row stride");
  static constexpr int kPad = 4;                                 // zero words in front of the planes: "row -1" / "row -2" of the first board
  static constexpr int kState = kPad;                            // [2][kNB4][RS]: black, white
  static constexpr int kMeta = kState + 2 * kNB4 * RS;           // flags[16], act[16], last[16], played[16], rng[32]
  static constexpr int kFair = kMeta + 6 * kNB4;                 // [16]: the progress words of this SIMD's wave slots (FairShare)
  static constexpr int kTmp = kFair + 16;                        // [2][2][RS]: layout change of one pair at load / store
  static constexpr int kUnion = kTmp + 4 * RS;
  // ply loop: per flood lane its result word (liberty class, size, role, seed) + the transpose buffer of the group masks
  static constexpr int kCls = kUnion;
  static constexpr int kSc = kCls + kWave;
  // per board FIVE blocks of RS words: one per flood lane of its quad + the flood of the mover's group G (all the lanes
  // that flood G write the same rows there).  The board stride in units of four words must be odd: phase 3 reads
  // twenty-five words per lane at this stride, and a stride of 4 x even words puts the sixteen boards on 2 or 4 banks
  static constexpr int kScPad = ((5 * RS / 4) % 2 == 0) ? 4 : 0;
  static constexpr int kScBoard = 5 * RS + kScPad;
  static constexpr int kLoopEnd = kSc + kNB4 * kScBoard;
  // load / store: the v2 analysis in its compact form (region 0 only: staging / transpose buffer); at store time the
  // emitter's scratch (2 x 128 words) and the spread table (uint2[256]); tracked boards: the parked mask / class rows
  static constexpr int kV2 = kUnion;
  static constexpr int kLut = kV2 + 256;
  static constexpr int kIoEnd = kV2 + (Lds2<R>::kRegion0 > 768 ? Lds2<R>::kRegion0 : 768);
  // byte-plane write-back of a whole group (emit_group): the group's 16 boards are ONE contiguous byte range, built as a
  // bit-string (bit i = byte i of the range) + the 8 bits -> 8 bytes table
  static constexpr int kGrpBits = kV2;
  static constexpr int kGrpWords = ((15 + kNB4 * 6 * R * R + 31) / 32 + 4) & ~3;
  static constexpr int kGrpLut = kGrpBits + kGrpWords;           // uint2[256]
  static constexpr int kGrpEnd = kGrpLut + 512;
  static constexpr int kTotal = (kLoopEnd > kIoEnd ? kLoopEnd : kIoEnd) > kGrpEnd ? (kLoopEnd > kIoEnd ? kLoopEnd : kIoEnd) : kGrpEnd;
  static_assert(kTotal * 4 <= 10240, "four waves per SIMD: 10 KB of LDS per wave");
  static_assert(kUnion % 4 == 0, "16-byte alignment of the flood blocks");
  static_assert(3 * kNB4 * RS <= kNB4 * kScBoard + kWave, "parked tracked rows fit the flood blocks");
};

// DPP inside a quad (quad_perm: lane i of each quad reads lane perm[i]); bound_ctrl off: every source lane exists
constexpr int QP_SHR1 = 0x90;   // [0,0,1,2]: lane i reads lane i-1 (lane 0 reads itself: mask it)
constexpr int QP_SHR2 = 0x40;   // [0,0,0,1]: lanes 2, 3 read lanes 0, 1
constexpr int QP_B0 = 0x00, QP_B3 = 0xFF;   // broadcast of lane 0 / lane 3
constexpr int QP_X1 = 0xB1, QP_X2 = 0x4E;   // [1,0,3,2] / [2,3,0,1]: butterfly
__device__ __forceinline__ uint32_t quad_or(uint32_t x) { x |= dpp0<QP_X1>(x); return x | dpp0<QP_X2>(x); }
__device__ __forceinline__ uint32_t quad_sum(uint32_t x) { x += dpp0<QP_X1>(x); return x + dpp0<QP_X2>(x); }

// The tt-th (0-based) set bit among the RPL row words v[] of a lane, p[] = their inclusive prefix counts (tt < p[RPL-1]):
// row index and bit position.  Branch-free on 0 / ~0 masks made by arithmetic shifts: a v_cmp + v_cndmask pair costs
// 8 issue cycles, sub + ashr + bitop3 costs 6 and needs no VCC round trip.
template <int RPL>
__device__ __forceinline__ void kth_set_bit(const uint32_t (&v)[RPL], const uint32_t (&p)[RPL], uint32_t tt, int &rr,
                                            uint32_t &pos) {
  uint32_t ntt = ~tt;
  uint32_t vr = v[0], base = 0, row = 0;
#pragma unroll
  for (int r = 1; r < RPL; ++r) {
    const uint32_t ge = (uint32_t)((int32_t)(p[r - 1] + ntt) >> 31);   // p[r-1] - 1 - tt < 0  <=>  tt >= p[r-1]
    vr = B3(ge, v[r], vr, T_SEL);
    base = B3(ge, p[r - 1], base, T_SEL);
    row -= ge;
  }
  ntt += base;                                                          // ~(tt - base)
  uint32_t ps = 0;
#pragma unroll
  for (int sh = 16; sh >= 1; sh >>= 1) {
    // c + ~tt = c - tt - 1 is negative iff tt >= c (the bit is in the upper half) - and is ~(tt - c) then: the popcount is added
    // to ~tt by the v_bcnt itself
    const uint32_t e = (uint32_t)__popc((vr >> ps) & ((1u << sh) - 1u)) + ntt;
    const uint32_t ge = (uint32_t)((int32_t)e >> 31);
    ntt = B3(ge, e, ntt, T_SEL);
    ps = B3((uint32_t)sh, ge, ps, T_ANDOR);                             // ps | (sh & ge)
  }
  rr = (int)row;
  pos = ps;
}

// 4-neighbourhood dilation of the RPL adjacent rows of a lane: the row above x[0] / below x[RPL-1] sits in the
// neighbouring lane (a non-existent row of a quad's last lane is zero in every set that is dilated, so nothing leaks
// from the previous board; what leaks into such a row from the next board is masked by the caller).  The centre point
// is not part of the result.
template <int RPL>
__device__ __forceinline__ void dilate_rows(const uint32_t (&x)[RPL], uint32_t (&d)[RPL]) {
  const uint32_t up = dpp0<0x138>(x[RPL - 1]), dn = dpp0<0x130>(x[0]);
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    const uint32_t above = r == 0 ? up : x[r - 1], below = r == RPL - 1 ? dn : x[r + 1];
    d[r] = B3(shl1(x[r]), x[r] >> 1, above, T_OR3) | below;
  }
}

// Byte-plane write-back of a whole group of the multi-ply kernel.  The boards of a group are one contiguous byte range
// [g, g + nbrd * 6 N^2): it is built as a bit-string in LDS - every quad ORs the rows of its board (stones from the LDS
// planes, the mask from its registers, the three uniform planes from the flag word) - and leaves as aligned 1 KB blocks,
// 64 lanes x 16 B through the 8 bits -> 8 bytes table: one hand-off instead of one per pair of boards, two ragged edges
// per group instead of two per board, and the store pattern that streams best (tools/ubench/write_patterns.hip).
// LPB = lanes per board (4: the quads of k_rollout4; 2: the pairs of k_rollout5, gg_v5.h), RPL rows per lane
template <int R, int RPL, int LPB = 4>
__device__ __forceinline__ void emit_group(uint8_t *g, int nbrd, int N, const uint32_t *st, int PL, int RS,
                                           const uint32_t (&inv_r)[RPL], const uint32_t *flagsv, uint32_t *bs, uint2 *lut,
                                           int lane) {
  const int P = N * N, S = 6 * P;
  const uint32_t mo = (uint32_t)((uintptr_t)g & 15u);
  const int nbits = (int)mo + nbrd * S;
  const int q4 = lane / LPB, r04 = RPL * (lane % LPB);
  WAVE_SYNC();
  for (int i = lane; i < (nbits + 31) / 32 + 1; i += kWave) bs[i] = 0;
  for (int e = lane; e < 256; e += kWave)
    lut[e] = make_uint2(__umul24((uint32_t)e & 15u, 0x204081u) & 0x01010101u, __umul24((uint32_t)e >> 4, 0x204081u) & 0x01010101u);
  WAVE_SYNC();
  if (q4 < nbrd) {
    const uint32_t fl = flagsv[q4];
    const uint32_t fullrow = (1u << N) - 1u;
    const uint32_t tp = (fl & 1u) ? fullrow : 0u, pp = (fl & 2u) ? fullrow : 0u, dp = (fl & 4u) ? fullrow : 0u;
    const uint32_t base = mo + (uint32_t)(q4 * S);
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const int rr = r04 + r;
      if (rr < N) {
        const uint32_t rows[6] = {st[0 * PL + q4 * RS + rr], st[1 * PL + q4 * RS + rr], tp, inv_r[r], pp, dp};
#pragma unroll
        for (int p = 0; p < 6; ++p) {
          if (rows[p]) {
            const uint32_t q = base + (uint32_t)(p * P + rr * N);
            const uint64_t x = (uint64_t)rows[p] << (q & 31u);
            atomicOr(bs + (q >> 5), (uint32_t)x);
            if ((uint32_t)(x >> 32)) atomicOr(bs + (q >> 5) + 1, (uint32_t)(x >> 32));
          }
        }
      }
    }
  }
  WAVE_SYNC();
  uint8_t *ga = g - mo;
  const int v0 = mo ? 1 : 0, v1 = nbits >> 4;
  const uint8_t *bb = reinterpret_cast<const uint8_t *>(bs);
  typedef uint32_t u4v __attribute__((ext_vector_type(4)));
  u4v *gv = reinterpret_cast<u4v *>(__builtin_assume_aligned(ga, 16));   // one global_store_dwordx4 per lane and round
  for (int v = v0 + lane; v < v1; v += kWave) {
    const uint2 lo = lut[bb[2 * v]], hi = lut[bb[2 * v + 1]];
    u4v o;
    o.x = lo.x; o.y = lo.y; o.z = hi.x; o.w = hi.y;
    gv[v] = o;
  }
  // the ragged vectors at either end of the group (shared with the neighbouring groups): single bytes
  const int head = mo ? 16 - (int)mo : 0, tail = nbits & 15;
  int j = -1;   // byte index inside the group
  if (lane < 16) { if (lane < head) j = lane; }
  else if (lane < 32 && lane - 16 < tail) j = nbrd * S - tail + (lane - 16);
  if (j >= 0 && j < nbrd * S) {
    const uint32_t q = mo + (uint32_t)j;
    g[j] = (uint8_t)((bs[q >> 5] >> (q & 31u)) & 1u);
  }
  WAVE_SYNC();
}

// cls word of a flood lane: bits 0-1 liberties of the group (saturated at 2), 2 the group is one stone, 3 the group
// exists (the seed was a stone), 4 the lane flooded G (the mover's group), 5 an opponent group without a liberty
// (captured); and what the lane knows about the board (the same in all four words of a quad, or zero): 6-7 liberties
// of G among the empty points (saturated at 2: G's own count, or the empty neighbours of q when the stone stands alone),
// 11 q has a friendly neighbour, 19 some on-board neighbour of q does NOT hold an opponent stone (clear: q is boxed in,
// state_utils.adj_data's `surrounded`, gym_go/state_utils.py:214-223).  (The seed of lane t is q's neighbour in direction t.)
constexpr uint32_t CL_LIBS = 3u, CL_ONE = 4u, CL_ANY = 8u, CL_G = 16u, CL_CAPT = 32u, CL_FRIEND = 1u << 11, CL_OPEN = 1u << 19;

// MOVES: the moves are given (moves: int32 [B][plies], gg_batch_play_moves) instead of drawn: a game stops at its first
// move that is out of range, on an invalid point or made after the game has ended; played_out[b] = moves applied.
// IO: 0 = byte planes (uint8 [B][6][N][N]), 1 = packed boards (uint32 [B][3N+1]), 2 = TRACKED boards (uint32 [B][5N+1]:
// the rows of black, white, invalid, multi_black, multi_white + the flag word - a packed board that carries its
// liberty classes, so that a launch needs no first analysis: per-ply stepping at the fused kernel's rate).
// IO == 3 (with MOVES, one ply): gogame.batch_next_states OUT OF PLACE on byte planes with a caller-owned WORKSPACE of
// tracked boards: `states` is the input batch (planes 0 / 1 and four flag bytes are read), env.states_out the output,
// env.ws holds the tracked form of whatever the previous call wrote for this game slot.  A board whose stones equal
// its workspace rows EXACTLY takes its liberty classes from there (a rollout that feeds every output back as the
// next input: every board, every call); any other board is analysed from scratch, so the result never depends on what
// the workspace holds as long as it was produced by this kernel or zero-filled.
// FULLN: the board fills the row capacity (N == R: 9, 13, 19) - the per-row "r < N" guards fold away at compile time.
// ENV (tracked boards, one ply): GoEnv.step for every game (gym_go/envs/go_env.py:49-76) - the action is given
// (MOVES, env.actions) or drawn; a finished game is reset first when auto_reset, refused otherwise; after the ply the
// kernel also writes status / dones / the action used / GoEnv.reward (:128-149, Tromp-Taylor areas on a rare path for
// the `real` method, every ply for `heuristic`) and, when asked, the byte-plane observation of every board.
struct EnvArgs {
  const int32_t *actions;   // int32 [B] or nullptr (MOVES instantiation only)
  float *rewards;           // each nullable
  uint8_t *dones;
  int32_t *status;
  int32_t *taken;
  uint8_t *states_out;      // uint8 [B][6][N][N]: the resulting position of EVERY game, or nullptr
  float komi;
  int heuristic;
  uint32_t *ws;             // IO == 3 only: the caller's workspace, uint32 [B][5N+1] (tracked boards of the last outputs)
  int canonical;            // IO == 3 only
  // WTS instantiation: the move of every game is DRAWN from these policy weights (float32 / bfloat16 / float16
  // [B][N*N+1], gg_ws.h: masked by the game's invalid-move rows, exact fixed-point inverse CDF, the game's generator)
  // instead of read from `actions` - gogame.random_weighted_action (gym_go/gogame.py:385-392) fused into the step
  const void *weights;
  int wdtype;               // GG_W_F32 / GG_W_BF16 / GG_W_F16
};

// WTS (ENV + MOVES): the move of every game is drawn from env.weights by the kernel (gg_ws.h) instead of read from env.actions
template <int R, int IO, bool MOVES = false, bool FULLN = false, bool ENV = false, bool WTS = false>
__global__ __launch_bounds__(kWave, 4) void k_rollout4(uint8_t *__restrict__ states, uint64_t *__restrict__ rng,
                                                       int32_t *__restrict__ last_actions, int64_t *__restrict__ steps_done,
                                                       int64_t B, int N, uint32_t inv, int plies, int auto_reset,
                                                       int nb, const int32_t *__restrict__ moves = nullptr,
                                                       int32_t *__restrict__ played_out = nullptr, EnvArgs env = EnvArgs()) {
  static_assert(!ENV || IO == 2, "the env step runs on tracked boards");
  static_assert(IO != 3 || (MOVES && !ENV), "the workspace step replays one given move per game");
  static_assert(!WTS || (ENV && MOVES), "policy-weighted moves belong to the env step's given-moves form");
  constexpr int RS = Lds4<R>::RS;
  constexpr int RV = (R + 3) / 4;
  constexpr int RPL = Lds4<R>::RPL;
  constexpr int PL = kNB4 * RS;   // words per plane of all boards
  constexpr int SCB = Lds4<R>::kScBoard;   // words per board in the flood blocks
  __shared__ __attribute__((aligned(16))) uint32_t lds[Lds4<R>::kTotal];
  if (FULLN) N = R;   // a compile-time constant from here on: row masks, r * N + c and the "row exists" tests fold
  uint32_t *st = lds + Lds4<R>::kState;     // st[colour * PL + board * RS + row]
  uint32_t *flagsv = lds + Lds4<R>::kMeta;  // bit 0 turn, 1 passed, 2 done, 3 on, 4 stopped / refused, 5 reset (dirty), 6 illegal move (IO 3)
  int *actv = reinterpret_cast<int *>(lds + Lds4<R>::kMeta + kNB4);
  int *lastv = reinterpret_cast<int *>(lds + Lds4<R>::kMeta + 2 * kNB4);
  int *playedv = reinterpret_cast<int *>(lds + Lds4<R>::kMeta + 3 * kNB4);
  uint32_t *rngv = lds + Lds4<R>::kMeta + 4 * kNB4;   // [2 * s], [2 * s + 1]
  uint32_t *tmp = lds + Lds4<R>::kTmp;      // tmp[(half * 2 + set) * RS + row], set 0 = invalid, 1 = M
  uint32_t *clsv = lds + Lds4<R>::kCls;
  uint32_t *sc = lds + Lds4<R>::kSc;
  uint32_t *v2 = lds + Lds4<R>::kV2;
  uint32_t *park = lds + Lds4<R>::kV2;      // tracked I/O: park[set * PL + board * RS + row], set 0 invalid, 1 mb, 2 mw
  uint2 *lut = reinterpret_cast<uint2 *>(lds + Lds4<R>::kLut);
  constexpr bool PACKED = IO == 1, TRACKED = IO == 2, CACHED = IO == 3;
  const int S = 6 * N * N, W = ((TRACKED || CACHED) ? 5 : 3) * N + 1;
  // nb (even, <= kNB4) boards per wave: the host picks it so that the groups fill the resident waves evenly
  const int64_t ngroups = (B + nb - 1) / nb;

  GG_WHERE_BEGIN;
  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int64_t b_first = g * nb;
    // The lane-derived values of the load phase are computed from a fresh lane id INSIDE the group loop (volatile asm:
    // not hoisted).  Computed once before the loop the compiler spills some of them across the first analysis and
    // reloads them one by one - each reload a scratch round trip behind a full `s_waitcnt vmcnt(0)` on the head of a
    // launch that, for the one-ply entry points, is one iteration long.
    int ln0;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln0));
    const Half hf = make_half(ln0, N, inv);
    const bool row = hf.hl < RS;
    const int q4 = hf.lane >> 2, t4 = hf.lane & 3, r04 = RPL * t4;   // outside the ply loop: board / first row of this lane
    // the next mover's invalid-move mask and the stones of groups with >= 2 liberties, rows r04 .. r04 + RPL - 1 of
    // board q4: in registers from here to the write-back
    uint32_t inv_r[RPL], M[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) inv_r[r] = M[r] = 0u;
    GG_PROF_DECL;
    // ---------------------------------------------------------------- load
    // (given moves: the first one is requested with the boards, so that it does not cost the first ply a round trip of its own;
    // the env step hands it back as the action taken)
    int mv_next = 0;
    if (MOVES && !WTS) mv_next = moves[((b_first + q4 < B) ? b_first + q4 : B - 1) * (int64_t)plies];
    const int mv_first = mv_next;
    WAVE_SYNC();
    if (TRACKED) {
      // the group's boards are ONE contiguous block of nb x (5 N + 1) words: a flat, fully coalesced copy (all loads in
      // flight at once); the stone planes go to their place, the mask / class rows are parked in the flood blocks and
      // picked up by their owner lanes
      const int64_t nbrd = (B - b_first) < nb ? (B - b_first) : nb;
      const int nw = (int)nbrd * W;
      const uint32_t *gp = reinterpret_cast<const uint32_t *>(states) + b_first * (int64_t)W;
      // Round 4: the block travels global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KB per instruction, no register
      // is held while it flies), ALL of it in flight at once, into the flood blocks (free until the first ply), and is
      // sorted from there: the stone rows into their planes, the flag words into the meta words, the mask / class rows
      // straight into the registers of their owner lanes.  (Round 3 fetched it through registers, eight loads per lane
      // at a time - all twenty-four at once cost 47 spilled registers - i.e. three dependent memory round trips on the
      // head of a launch that, for the one-ply entry points, is one iteration long.)
      constexpr int KD = ((kNB4 * (5 * R + 1) * 4 + 12 + 15) / 16 + kWave - 1) / kWave;   // DMA instructions per lane
      static_assert(((kNB4 * (5 * R + 1) * 4 + 12 + 15) / 16) * 4 <= kNB4 * Lds4<R>::kScBoard, "a group's tracked block fits the flood blocks");
      const uint8_t *gb = reinterpret_cast<const uint8_t *>(gp);
      const uint32_t mis = (uint32_t)((uintptr_t)gb & 15u);   // (a multiple of 4: the aligned superset of the block is fetched)
      const int nvec = (int)((mis + (uint32_t)nw * 4u + 15u) >> 4);
      WAVE_SYNC();
      lds_drain();   // (whatever read the flood blocks before - the previous group's write-back - has its data: the DMA's LDS writes are not ordered with the DS queue)
      {
        const uint32_t stage_lds = lds_addr(sc);
#pragma unroll
        for (int k = 0; k < KD; ++k) {
          const int v = hf.lane + kWave * k;
          if (v < nvec) dma16(gb - mis + 16 * v, stage_lds + 1024u * (uint32_t)k);   // lane L's 16 bytes land at m0 + 16 L
        }
      }
      // the generator states of the group: requested with the block, so that they do not cost a round trip of their own
      uint64_t xg = 0;
      if ((!MOVES || WTS) && hf.lane < kNB4) xg = rng[(hf.lane < nb && b_first + hf.lane < B) ? b_first + hf.lane : B - 1];
      for (int i = hf.lane; i < 2 * PL; i += kWave) st[i] = 0;      // rows N .. RS-1 and absent boards read as zero
      if (hf.lane < Lds4<R>::kPad) lds[hf.lane] = 0;
      dma_wait();
      WAVE_SYNC();
      const uint32_t *stg = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(sc) + mis);   // word i of the block
      {
        const uint32_t N2 = 2u * (uint32_t)N, inv2 = (65536u + N2 - 1u) / N2;   // i / (2 N) exactly for i < 16 * 2 N
        for (int i = hf.lane; i < (int)nbrd * (int)N2; i += kWave) {
          const int sb = (int)(((uint32_t)i * inv2) >> 16), w = i - sb * (int)N2;
          const int pl = w >= N ? 1 : 0;
          st[pl * PL + sb * RS + (w - pl * N)] = stg[sb * W + w];
        }
      }
      if (hf.lane < kNB4) {
        const int sb = hf.lane;
        const bool on = sb < nb && b_first + sb < B;
        flagsv[sb] = on ? ((stg[sb * W + 5 * N] & 7u) | 8u) : 0u;
        lastv[sb] = -1;
        playedv[sb] = 0;
        if (!MOVES || WTS) {
          rngv[2 * sb] = (uint32_t)xg;
          rngv[2 * sb + 1] = (uint32_t)(xg >> 32);
        }
      }
      {
        const bool have = q4 < (int)nbrd;
        const uint32_t *bq = stg + (have ? q4 : 0) * W;
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          const int rw = r04 + r;
          const bool ok = have && rw < N;
          const int rc = ok ? rw : 0;
          const uint32_t iv = bq[2 * N + rc], mb = bq[3 * N + rc], mw = bq[4 * N + rc];
          inv_r[r] = ok ? iv : 0u;
          M[r] = ok ? (mb | mw) : 0u;
        }
      }
      WAVE_SYNC();
    } else {
      if (hf.lane < kNB4) flagsv[hf.lane] = 0;                       // boards beyond nb: off
      for (int i = hf.lane; i < 2 * PL; i += kWave) st[i] = 0;
      if (hf.lane < Lds4<R>::kPad) lds[hf.lane] = 0;
      WAVE_SYNC();
    }
    // Byte planes / packed boards: pairs, first classes by the per-ply-style analysis.  The global loads of pair i + 1 are
    // issued (into registers) before pair i is converted: a launch of few plies is a latency chain, and eight
    // dependent load -> LDS -> convert rounds in a row were a third of a one-ply launch.
    constexpr int NVL = CACHED ? (2 * R * R + 15 + 15) / 512 + 1 : (4 * R * R + 15 + 15) / 512 + 1;   // 16-byte vectors per lane
    static_assert(NVL <= 3, "vectors per lane of a staged board");
    if (CACHED) {   // the actions of the whole group first: the address of a board's INVD[action] byte depends on them
      if (hf.lane < kNB4) actv[hf.lane] = (b_first + hf.lane < B) ? moves[b_first + hf.lane] : 0;
      WAVE_SYNC();
    }
    if (!MOVES && !TRACKED && hf.lane < kNB4) {   // the generator states of the whole group: one coalesced load, not one per pair
      const uint64_t x = rng[(b_first + hf.lane < B) ? b_first + hf.lane : B - 1];
      rngv[2 * hf.lane] = (uint32_t)x;
      rngv[2 * hf.lane + 1] = (uint32_t)(x >> 32);
    }
    // (plain scalars, not a struct: the prefetched values must stay in registers)
    uint4 cv0 = make_uint4(0, 0, 0, 0), cv1 = cv0, cv2 = cv0, nv0 = cv0, nv1 = cv0, nv2 = cv0;
    uint32_t cfb = 0, cwb = 0, cww = 0, cwm = 0, nfb = 0, nwb = 0, nww = 0, nwm = 0;
    int ca = 0, na = 0;
#define GG_ISSUE_PAIR(I, V0, V1, V2, FB, WB, WW, WM, A)                                                               \
    do {                                                                                                               \
      const int s_ = 2 * (I) + hf.h;                                                                                   \
      const int64_t b_ = (b_first + s_ < B) ? b_first + s_ : B - 1;                                                    \
      if (!PACKED) {                                                                                                   \
        const uint8_t *gs_ = states + b_ * (int64_t)S;                                                                 \
        int pt_ = 0;                                                                                                   \
        if (CACHED) {                                                                                                  \
          A = actv[s_];                                                                                                \
          pt_ = (A >= 0 && A < hf.P) ? A : 0;                                                                          \
          const uint32_t *wsb_ = env.ws + b_ * (int64_t)W;                                                             \
          if (hf.hl < N) { WB = wsb_[hf.hl]; WW = wsb_[N + hf.hl]; WM = wsb_[3 * N + hf.hl] | wsb_[4 * N + hf.hl]; }   \
        }                                                                                                              \
        FB = 0;                                                                                                        \
        if (hf.hl < 4) {                                                                                               \
          const int off_ = hf.hl == 0 ? 2 * hf.P : hf.hl == 1 ? 3 * hf.P + pt_ : hf.hl == 2 ? 4 * hf.P : 5 * hf.P;     \
          FB = gs_[off_];                                                                                              \
        }                                                                                                              \
        const uint32_t mis_ = (uint32_t)((uintptr_t)gs_ & 15u);                                                        \
        const uint4 *ga_ = reinterpret_cast<const uint4 *>(gs_ - mis_);                                                \
        const int nv_ = (int)(mis_ + (CACHED ? 2 : 4) * hf.P + 15) >> 4;                                               \
        if (hf.hl < nv_) V0 = ga_[hf.hl];                                                                              \
        if (NVL > 1 && hf.hl + 32 < nv_) V1 = ga_[hf.hl + 32];                                                         \
        if (NVL > 2 && hf.hl + 64 < nv_) V2 = ga_[hf.hl + 64];                                                         \
      }                                                                                                                \
    } while (0)
    if (!TRACKED && nb >= 2) GG_ISSUE_PAIR(0, cv0, cv1, cv2, cfb, cwb, cww, cwm, ca);
#pragma unroll 1
    for (int i = 0; i < (TRACKED ? 0 : nb / 2); ++i) {
      if (i + 1 < nb / 2) GG_ISSUE_PAIR(i + 1, nv0, nv1, nv2, nfb, nwb, nww, nwm, na);
      const int s = 2 * i + hf.h;
      const bool on = b_first + s < B;
      const int64_t b = on ? b_first + s : B - 1;
      uint32_t black, white, invalid, mb = 0, mw = 0;
      int turn, passed, done;
      bool illegal = false;   // IO == 3: the given move is out of range or on a set point of plane 3
      if (PACKED) {
        uint32_t fw;
        load_packed_h(reinterpret_cast<const uint32_t *>(states) + b * (int64_t)W, N, hf, black, white, invalid, fw);
        turn = fw & 1u; passed = (fw >> 1) & 1u; done = (fw >> 2) & 1u;
      } else {
        const uint8_t *gs = states + b * (int64_t)S;
        uint8_t *io = reinterpret_cast<uint8_t *>(v2) + hf.h * Cfg<R>::kIoBytes;
        const uint32_t mi = (uint32_t)((uintptr_t)gs & 15u);
        const int nv = (int)(mi + (CACHED ? 2 : 4) * hf.P + 15) >> 4;
        const uint32_t flags = half_of(__ballot(cfb != 0), hf.h) & 0xFu;   // bit 0 turn, 1 INVD[pt], 2 passed, 3 done
        WAVE_SYNC();
        uint4 *iov = reinterpret_cast<uint4 *>(io);
        if (hf.hl < nv) iov[hf.hl] = cv0;
        if (NVL > 1 && hf.hl + 32 < nv) iov[hf.hl + 32] = cv1;
        if (NVL > 2 && hf.hl + 64 < nv) iov[hf.hl + 64] = cv2;
        WAVE_SYNC();
        black = plane_to_row<R>(io + mi, N, hf.hl);
        white = plane_to_row<R>(io + mi + hf.P, N, hf.hl);
        turn = flags & 1u; passed = (flags >> 2) & 1u; done = (flags >> 3) & 1u;
        if (CACHED) {
          // the classes come from the workspace when its stones are this board's stones (planes 0 / 1, exactly)
          invalid = 0;
          const bool in_range = ca >= 0 && ca <= hf.P;
          illegal = !in_range || (ca < hf.P && (flags & 2u));   // gogame.py:59
          const bool miss = half_of(__ballot(black != cwb || white != cww), hf.h) != 0u;
          if (__ballot(on && miss)) {   // some board of the pair is new to the workspace: analyse (both halves run)
            uint32_t ab;
            analyze2<R, false>(black, white, hf.full_l1 & ~(black | white), hf, v2, mb, ab, mw, nullptr, nullptr, true);
          }
          if (!miss) { mb = cwm; mw = 0; }
        } else {
          invalid = plane_to_row<R>(io + mi + 3 * hf.P, N, hf.hl);
        }
      }
      if (!CACHED) {
        uint32_t ab;
        analyze2<R, false>(black, white, hf.full_l1 & ~(black | white), hf, v2, mb, ab, mw, nullptr, nullptr, true);
      }
      if (row) {
        st[0 * PL + s * RS + hf.hl] = black;
        st[1 * PL + s * RS + hf.hl] = white;
        tmp[(hf.h * 2 + 0) * RS + hf.hl] = invalid;
        tmp[(hf.h * 2 + 1) * RS + hf.hl] = mb | mw;
      }
      if (hf.hl == 0) {
        flagsv[s] = (uint32_t)turn | ((uint32_t)passed << 1) | ((uint32_t)done << 2) | (on ? 8u : 0u) | (illegal ? 64u : 0u);
        lastv[s] = -1;
        playedv[s] = 0;
      }
      WAVE_SYNC();
      if ((hf.lane >> 3) == i) {   // the two quads that own this pair pick their rows up
        const uint32_t *tp = tmp + ((q4 & 1) * 2) * RS + r04;
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          inv_r[r] = tp[r];
          M[r] = tp[RS + r];
        }
      }
      WAVE_SYNC();
      cv0 = nv0; cv1 = nv1; cv2 = nv2; cfb = nfb; cwb = nwb; cww = nww; cwm = nwm; ca = na;
    }
#undef GG_ISSUE_PAIR

    // ---------------------------------------------------------------- policy-weighted moves (ENV + MOVES + weights)
    int *wact = reinterpret_cast<int *>(tmp);   // [16] the drawn moves (tmp is free on tracked boards)
    if (WTS) {
      constexpr int NJ = (R * R + 1 + 15) / 16, AMIN = FULLN ? R * R + 1 : 5;
      // every lane-derived value of this block comes from a FRESH lane id (volatile asm: neither hoisted nor merged).
      // Derived from the ids computed before the group loop they are spilled across it, and each reload is a scratch
      // round trip that also waits for the weight loads in flight: 25 dependent round trips, +12 us on the launch.
      int lw;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lw));
      const int q4w = lw >> 2, t4w = lw & 3, r04w = RPL * t4w, rw = lw >> 4;
      const int Pw = N * N, A = Pw + 1;
      uint32_t *vb = sc;        // [16][16]: the boards' valid-action bit-strings (the flood blocks are free before the ply)
      uint32_t *uh = clsv;      // [16] draws, [16 .. 31] "this game draws"
      WAVE_SYNC();
      for (int i = lw; i < kNB4 * 16; i += kWave) vb[i] = 0;
      if (lw < kNB4) {
        const uint32_t fl = flagsv[lw];
        const bool draws = ((fl >> 3) & 1u) && !(((fl >> 2) & 1u) && !auto_reset);   // a frozen game keeps its generator
        uint32_t hi = 0;
        if (draws) {
          uint64_t x = ((uint64_t)rngv[2 * lw + 1] << 32) | rngv[2 * lw];
          hi = (uint32_t)(splitmix_next(x) >> 32);
          rngv[2 * lw] = (uint32_t)x;
          rngv[2 * lw + 1] = (uint32_t)(x >> 32);
        }
        uh[lw] = hi;
        uh[kNB4 + lw] = draws ? 1u : 0u;
        wact[lw] = -1;
      }
      // the weights of the first four boards: in flight while the bit-strings are built
      uint32_t nbits[NJ];
      wload_row<NJ, AMIN>(env.weights, (b_first + rw < B) ? b_first + rw : B - 1, env.wdtype, A, lw, nbits);
      WAVE_SYNC();
      {   // every quad ORs the playable points of its rows into its board's string (a game being reset plays on the empty board)
        const uint32_t fl = flagsv[q4w];
        const bool resets = ((fl >> 2) & 1u) && auto_reset;
        const uint32_t fullrow = (1u << N) - 1u;
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          const int rr = r04w + r;
          if (rr < N) {
            const uint32_t ok = resets ? fullrow : (fullrow & ~inv_r[r]);
            const uint32_t q = (uint32_t)(rr * N);
            const uint64_t sh = (uint64_t)ok << (q & 31u);
            atomicOr(vb + 16 * q4w + (q >> 5), (uint32_t)sh);
            if ((uint32_t)(sh >> 32)) atomicOr(vb + 16 * q4w + (q >> 5) + 1, (uint32_t)(sh >> 32));
          }
        }
        if (t4w == 0) atomicOr(vb + 16 * q4w + (Pw >> 5), 1u << (Pw & 31));   // the pass
      }
      WAVE_SYNC();
      // four boards per pass, one DPP row each; the weights of the next pass are in flight while this one is drawn
#pragma unroll 1
      for (int ps = 0; ps < 4; ++ps) {
        uint32_t bits[NJ], vm[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bits[j] = nbits[j];
        const int s = 4 * ps + rw;
        if (ps + 1 < 4) {
          const int s1 = s + 4;
          wload_row<NJ, AMIN>(env.weights, (b_first + s1 < B) ? b_first + s1 : B - 1, env.wdtype, A, lw, nbits);
        }
        wmask_from_bits<NJ>(vb + 16 * s, lw, vm);
        const int a = wsample_row<NJ>(bits, vm, uh[s], lw);
        if (a != -2 && s < nb && b_first + s < B && uh[kNB4 + s] != 0u) wact[s] = a;
      }
      WAVE_SYNC();
    }

    // ---------------------------------------------------------------- the plies
    GG_PROF(6);   // load
    FairShare fair(lds + Lds4<R>::kFair);
    // (a wide band - an eighth of the launch, at most 24 plies: measured 4 / 8 / 16 / 24 / 32 plies at 256 plies per launch:
    // 2.076 / 2.062 / 2.049 / 2.042 / 2.046 ms)
#ifndef GG_AB_FAIRLAG
#define GG_AB_FAIRLAG 24u
#endif
#ifndef GG_AB_FAIRSTEP
#define GG_AB_FAIRSTEP 3
#endif
#ifndef GG_AB_EARLY4
#define GG_AB_EARLY4 false
#endif
    const uint32_t fair_lag = plies >= 192 ? GG_AB_FAIRLAG : (plies >= 16 ? (uint32_t)plies >> 3 : 2u);
    uint32_t uq = 0;   // (drawn moves) this lane's pre-mixed draw: lane j of a quad holds the one of ply (t & ~3) + j, rotated by one lane per ply
#pragma unroll 1
    for (int t = 0; t < plies; ++t) {
      // fair share of the SIMD (gg_common.h): every fourth ply the wave publishes the ply it has reached and sets its issue
      // priority by how many of its SIMD-mates are >= fair_lag plies behind it
      if ((t & GG_AB_FAIRSTEP) == 0 && plies >= 8) {
        // ... and the band never exceeds the plies that are left: towards the end of the launch the stragglers are let
        // through, the four waves of a SIMD reach their write-back together (2.046 -> 2.022 ms per 256-ply launch;
        // half / a quarter / an eighth of the plies left: 2.027 / 2.032 / 2.040; checks every 2 / 8 plies: 2.05 / 2.09)
        const uint32_t left = (uint32_t)(plies - t);
        fair.update((uint32_t)t, left < fair_lag ? (left > 2u ? left : 2u) : fair_lag);
      }
      // the lane-derived indices of the three phases are recomputed every ply (a few VALU ops) instead of being hoisted
      // out of the loop, where they end up in scratch: a reload is a vector-memory round trip at the top of each phase
      // (volatile asm: neither hoisted nor merged)
      int ln;   // = hf.lane (one wave per workgroup), straight from the hardware
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      const int s4 = (ln >> 2) & 15, t5 = ln & 3, r0 = RPL * t5;   // board, lane of the quad, first row of this lane
      const bool bl = s4 < nb;
      uint32_t full[RPL];   // the N-bit row mask of the lane's rows that exist
#pragma unroll
      for (int r = 0; r < RPL; ++r) full[r] = (r0 + r < N) ? (1u << N) - 1u : 0u;

      // the board's move of this ply and its flag word as the ply finds it (after an auto-reset), the same in the four
      // lanes of a quad: the three phases run on one lane assignment, so they travel in registers - read back from LDS
      // they cost phases 2 and 3 a dependent round trip each (action -> addresses of the rows around it -> rows)
      int a_q;
      uint32_t fl_q;
      // phase 1 - four lanes per board, RPL rows each: liveness, the generator (drawn redundantly by the four lanes),
      // the k-th valid point of the mask (or the given move)
      {
        const uint32_t fl = flagsv[s4];
        const bool on = bl && ((fl >> 3) & 1u);
        const bool done = (fl >> 2) & 1u;
        bool live, reset, place = false, wr_act;
        int rabs = 0, a;
        uint32_t pos = 0;
        uint64_t x = 0;
        if (MOVES) {
          // the move of this ply was fetched during the previous one (mv_next), the next one is requested now
          const int64_t bm = (b_first + s4 < B) ? b_first + s4 : B - 1;
          constexpr bool drawn = WTS;   // the move was drawn from the policy weights above
          const int mv = drawn ? wact[s4] : mv_next;
          if (!drawn && t + 1 < plies) mv_next = moves[bm * (int64_t)plies + t + 1];
          // (gg_batch_play_moves passes auto_reset = 0: a finished game stops; the env step may reset it first, and the
          // reset stands even when the move is then refused - GoEnv.reset comes before the action check)
          reset = !CACHED && on && done && auto_reset != 0 && !((fl >> 4) & 1u);
          // (gogame.next_state itself does not look at the game-over plane: the workspace step plays on, like the reference)
          live = on && (CACHED || !done || reset) && !((fl >> 4) & 1u) && mv >= 0 && mv <= hf.P && !(CACHED && ((fl >> 6) & 1u));
          a = hf.P;
          const bool pt = live && mv < hf.P;
          int ar = 0, ac = 0;
          if (pt) split_action(mv, N, hf.inv, ar, ac);
          // the lane that owns row ar tests the mask bit, the quad shares the verdict (a board being reset is empty)
          uint32_t bad = 0;
#pragma unroll
          for (int r = 0; r < RPL; ++r) bad |= (!CACHED && pt && !reset && r0 + r == ar) ? ((inv_r[r] >> ac) & 1u) : 0u;
          const bool illegal = quad_or(bad) != 0u;
          if (pt) {
            live = !illegal;
            rabs = ar; pos = (uint32_t)ac;
            a = mv;
          }
          wr_act = bl && t5 == 0;
          if (on && !live && wr_act) flagsv[s4] = fl | 16u;   // stopped for good
          if (!live) a = -1;
          place = wr_act && a >= 0 && a < hf.P;
        } else {
          live = on && !(done && !auto_reset);
          reset = live && done;           // auto-reset: the board is init_state from now on
          uint32_t v[RPL], p[RPL];
          const uint32_t rm = reset ? ~0u : 0u;   // a board being reset plays on the empty board
#pragma unroll
          for (int r = 0; r < RPL; ++r) {
            v[r] = B3(full[r], rm, inv_r[r], TA & (TB | (~TC & 0xFF)));   // full: 0 for rows >= N
            p[r] = (uint32_t)__popc(v[r]) + (r ? p[r - 1] : 0u);
          }
          const uint32_t T = p[RPL - 1];
          // valid points of the board up to and including this lane (Sx) and on the whole board (n): quad scan
          // (the DPP moves are evaluated by every lane, THEN masked: inside a conditional the source lanes would be off)
          // (a DPP bank mask cannot do the masking: its banks are the four QUADS of a row, not the lanes of a quad)
          const uint32_t sh1 = dpp0<QP_SHR1>(T);
          const uint32_t x1 = T + (t5 >= 1 ? sh1 : 0u);
          const uint32_t sh2 = dpp0<QP_SHR2>(x1);
          const uint32_t Sx = x1 + (t5 >= 2 ? sh2 : 0u);
          const uint32_t n = dpp0<QP_B3>(Sx), P = Sx - T;
          // The draws of a board, FOUR plies at a time: the generator is a counter (x += c per draw) and a board draws once per
          // ply from ply 0 until it freezes for good, so the draw of ply t is mix(x0 + (t + 1) c) with x0 the generator the launch
          // found (it stays in LDS untouched until the write-back, which leaves x0 + played c).  Lane j of the quad mixes the
          // draw of ply t + j every fourth ply; a ply takes lane 0's and the quad rotates by one lane: 2 DPP moves instead of
          // the 64-bit mix (19 VALU instructions, six of them multiplies) + an LDS round trip of the generator per ply.
          if ((t & 3) == 0) {
            uint64_t xx = (((uint64_t)rngv[2 * s4 + 1] << 32) | rngv[2 * s4]) + (uint64_t)(uint32_t)(t + t5) * 0x9E3779B97F4A7C15ull;
            uq = (uint32_t)(splitmix_next(xx) >> 32);
          }
          const uint32_t uh = dpp0<QP_B0>(uq);
          uq = dpp0<0x39>(uq);   // quad_perm [1,2,3,0]: lane i takes lane i + 1's
          const uint32_t k = __umulhi(uh, n + 1u);   // k == n: the pass
          const bool hit = k >= P && k < P + T;        // this lane holds the k-th valid point
          int rr;
          kth_set_bit<RPL>(v, p, (k - P) & 0x3FFu, rr, pos);   // (only the hit lane's result is used; the mask keeps tt small elsewhere)
          rabs = r0 + rr;
          // -1: the board does not move this ply.  The lane with the point announces it; a pass / an idle board is
          // announced by the board's first lane
          a = !live ? -1 : (k < n ? rabs * N + (int)pos : hf.P);
          wr_act = bl && (live && k < n ? hit : t5 == 0);
          place = bl && live && hit;
          if (ENV && on && !live && t5 == 0) flagsv[s4] = fl | 16u;   // a frozen game refuses the step
        }
        // (one lane of a board announces the move - the one that holds the point, or the first: a quad OR hands it round)
        a_q = MOVES ? a : (int)quad_or(wr_act ? (uint32_t)(a + 2) : 0u) - 2;
        fl_q = reset ? 40u : fl;   // a board being reset: on, dirty, black to move
        // (a finished game whose move is refused is still reset when auto_reset - GoEnv.reset comes before the action
        // check - also when no board of the wave moves: the resets are applied before the early exit)
        uint64_t resetm = __ballot(reset && t5 == 0);
        const bool none_live = __ballot(live) == 0;
        if (none_live && resetm == 0) break;
        if (resetm) {   // rare
          if (reset) {
#pragma unroll
            for (int r = 0; r < RPL; ++r) inv_r[r] = M[r] = 0u;
          }
          while (resetm) {
            const int s = (__ffsll((unsigned long long)resetm) - 1) >> 2;   // lane 4 s -> board s
            resetm &= resetm - 1;
            for (int i = hf.lane; i < 2 * RS; i += kWave) st[(i / RS) * PL + s * RS + (i % RS)] = 0;
            if (hf.lane == 0) flagsv[s] = (flagsv[s] & 16u) | 8u | 32u;   // on, reset (written back even if nothing is played)
          }
          if (none_live) break;
        }
        WAVE_SYNC();
        // the new stone goes into the mover's plane right away: every later phase sees the position with it
        // ... and, as the group G it forms on its own, into the board's G block (phase 2 floods over it when q has a
        // friendly neighbour): phase 3 then reads G from one place whatever the case
        if (place) {
          const int turn = reset ? 0 : (int)(fl & 1u);
          st[turn * PL + s4 * RS + rabs] |= 1u << pos;
          uint32_t *gb = sc + s4 * SCB + 4 * RS;
          uint4 *pz = reinterpret_cast<uint4 *>(gb);
#pragma unroll
          for (int i = 0; i < RV; ++i) pz[i] = make_uint4(0u, 0u, 0u, 0u);
          asm volatile("" ::: "memory");
          gb[rabs] = 1u << pos;
        }
      }
      WAVE_SYNC();
      GG_PROF(0);

      // phase 2 - one lane per (board, direction): lane t of a quad looks at ONE neighbour of q (0 up, 1 down, 2 left, 3 right).
      // An opponent stone there: the lane floods that stone's group.  A friendly stone: the lane floods G from q (every such
      // lane of the board runs the same flood and writes the same rows into the board's G block, a benign duplicate - the
      // flood batch costs the same whatever its lanes carry - so no lane has to find out which of them is "the" G lane).
      // Nothing there: the lane idles (an all-zero flood).  What the board-level tests of phase 3 need - empty neighbours
      // of q, is any of them friendly, is q boxed in - is one packed quad sum, and travels in the lanes' class words.
      // (round 3 compacted the opponent neighbours onto lanes 0.. and kept G on lane 3: 128 VALU instructions of
      // compares and selects per wave-ply for the role assignment alone, against 45 here.)
      {
        const int a = a_q;
        const uint32_t turn = fl_q & 1u;
        // this lane's flood block is cleared first: the flood's seed row set is staged through it (below), and a lane
        // that floods G or nothing must leave it empty for phase 3
        uint32_t *blk = sc + s4 * SCB + t5 * RS;
        {
          uint4 *pz = reinterpret_cast<uint4 *>(blk);
#pragma unroll
          for (int i = 0; i < RV; ++i) pz[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        const uint32_t mv1 = ((uint32_t)a < (uint32_t)hf.P) ? 1u : 0u;   // a stone was placed (not a pass, not an idle board)
        int ar, ac;
        split_action(mv1 ? a : 0, N, hf.inv, ar, ac);
        const int sg = 2 * (t5 & 1) - 1;
        const int dr = (t5 & 2) ? 0 : sg, dc = (t5 & 2) ? sg : 0;
        const int nr = ar + dr, nc = ac + dc;   // this lane's neighbour of q: row -1 .. N, column -1 .. N
        const uint32_t *pm = st + turn * PL + s4 * RS, *po = st + (1u - turn) * PL + s4 * RS;
        // (row -1 / -2 of a board is the zero row 19 / 18.. of the board before it or the pad in front of the planes, row N
        // a zero row, row N + 1 - read for an off-board neighbour only - whatever follows: masked by obit)
        const uint32_t rowm = pm[nr], rowo = po[nr], oup = po[nr - 1], odn = po[nr + 1];
        const uint32_t ncs = (uint32_t)nc & 31u;   // column -1 reads bit 31, column N bit N: never set in a row
        const uint32_t mbit = (rowm >> ncs) & mv1, obit = (rowo >> ncs) & mv1;
        const uint32_t onb = ((uint32_t)nr < (uint32_t)N && (uint32_t)nc < (uint32_t)N) ? mv1 : 0u;
        const uint32_t ex = mbit | obit;           // this lane floods
        // quad totals in one packed word: bits 0-2 empty neighbours of q, 8-10 friendly ones, 16-18 on-board neighbours
        // that are not the opponent's (none: q is boxed in, state_utils.adj_data's `surrounded`)
        // (+ 7 in a three-bit field carries into the bit above it iff the field is not zero: bit 11 = q has a friendly
        // neighbour, bit 19 = q is NOT boxed in)
        const uint32_t qs = quad_sum((onb & ~ex) | (mbit << 8) | ((onb & ~obit) << 16)) + 0x70700u;
        const uint32_t ne = qs & 7u, ne2 = ne < 2u ? ne : 2u;
        const uint32_t lone = ~(uint32_t)__builtin_amdgcn_sbfe((int)qs, 11, 1);   // ~0: no friendly neighbour
        // the seed: q itself for G, the neighbour stone for an opponent group
        const uint32_t gm = 0u - mbit;
        const int sr = nr - (dr & (int)gm), scol = nc - (dc & (int)gm);
        // the colour this lane floods (G: the mover's) and the other one, as 16-byte row sets (indexed from the aligned
        // base of the LDS array: behind the pad the compiler no longer sees the alignment of `st + ...` and would split the
        // row loads into dwords)
        const uint32_t ownc = turn ^ mbit ^ 1u;
        const uint4 *lds4 = reinterpret_cast<const uint4 *>(lds);
        const uint4 *pmv = lds4 + (Lds4<R>::kState + (int)ownc * PL + s4 * RS) / 4;
        const uint4 *pov = lds4 + (Lds4<R>::kState + (int)(ownc ^ 1u) * PL + s4 * RS) / 4;
        uint32_t *out = mbit ? sc + s4 * SCB + 4 * RS : blk;
        // the opponent's stone is a group of its own iff none of its neighbours holds an opponent stone (ko needs it)
        const uint32_t onbr = B3(oup, odn, rowo >> 1, T_OR3) | shl1(rowo);
        const uint32_t single = obit & ~(onbr >> ncs);
        // the class word (below) but for the liberties of the lane's group, which the flood has yet to find
        const uint32_t pre = (qs & (CL_FRIEND | CL_OPEN)) | (single << 2) | (ex << 3) | (mbit << 4) | ((ne2 & lone) << 6);
        uint32_t cnt = 0;
        {
          uint32_t m[R];
          {
            uint32_t mrev[R], f[R];
            uint32_t mt[RV * 4], ft[RV * 4];
#pragma unroll
            for (int i = 0; i < RV; ++i) {
              const uint4 x = pmv[i];
              mt[4 * i] = x.x; mt[4 * i + 1] = x.y; mt[4 * i + 2] = x.z; mt[4 * i + 3] = x.w;
            }
            // the seeds are one bit of one row: written into the cleared block at its (dynamic) row and read back as the
            // flood's row set - two LDS instructions instead of a select per row (the flood keeps its odd rows
            // bit-reversed; a lane without a seed writes a zero to row 0); the block is cleared again behind the read
            {
              const int srw = sr & (int)(0u - ex);
              asm volatile("" ::: "memory");   // (DS instructions of a wave execute in order; the compiler must keep it too)
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
            }
            // (measured on this kernel, 65 536 games x 256 plies: the two-chain flood2_dual 2.78 ms against 2.33 ms, a first
            // closure test already after the second sweep 2.43 ms)
            GG_PROF(1);
            // (the closure test's copy of the fill goes to LDS four rows per ds_write_b128: 32-bit stores of one row
            // from lanes RS = 20 words apart are a 4-way bank conflict - removing it measured 2.327 vs 2.327 ms per
            // 256-ply launch: the LDS is not on this kernel's critical path)
            flood2_serial<R, true, true, GG_AB_EARLY4>(m, mrev, f, out);
            GG_PROF(2);
          }
          // liberties of this lane's group on the position with the new stone (captures not yet removed); m[] still
          // holds the flooded colour's rows
          uint32_t gt[RV * 4], ot[RV * 4];
          const uint4 *pg = reinterpret_cast<const uint4 *>(out);
#pragma unroll
          for (int i = 0; i < RV; ++i) {
            const uint4 x = pg[i], y = pov[i];
            gt[4 * i] = x.x; gt[4 * i + 1] = x.y; gt[4 * i + 2] = x.z; gt[4 * i + 3] = x.w;
            ot[4 * i] = y.x; ot[4 * i + 1] = y.y; ot[4 * i + 2] = y.z; ot[4 * i + 3] = y.w;
          }
          const uint32_t fullrow = (1u << N) - 1u;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const uint32_t e = (FULLN || r < N) ? B3(ot[r], m[r], fullrow, ~(TA | TB) & TC & 0xFF) : 0u;   // empty points
            const uint32_t up = r > 0 ? gt[r - 1] : 0u, dn = r + 1 < R ? gt[r + 1] : 0u;   // rows >= R are not written
            const uint32_t dd = B3(shl1(gt[r]), gt[r] >> 1, up, T_OR3);
            const uint32_t l = B3(dd, dn, e, (TA | TB) & TC);
            cnt += (uint32_t)__popc(l);   // only min(cnt, 2) is used: one accumulating v_bcnt per row
          }
        }
        // liberties of G among the empty points, as phase 3 wants them (bits 6-7; phase 3 ORs the four words of a quad):
        // G's own count from the lanes that flooded it, the empty neighbours of q (in `pre`) when the stone stands alone
        const uint32_t lib2 = cnt < 2u ? cnt : 2u;
        const uint32_t gsel6 = (uint32_t)__builtin_amdgcn_sbfe((int)pre, 4, 1);   // ~0: this lane flooded G
        const uint32_t pre2 = pre + pre;   // CL_ANY at bit 4, CL_G at bit 5; doubled again: CL_ANY at bit 5
        const uint32_t dead = ((cnt - 1u) >> 26) & B3(pre2 + pre2, pre2, CL_CAPT, TA & ~TB & TC & 0xFF);   // no liberty, flooded, not G
        clsv[ln] = B3(lib2 << 6, gsel6, pre, T_ANDOR) | lib2 | dead;
        // an opponent group that keeps >= 2 liberties keeps its class: phase 3 must not see it
        if (!mbit && cnt >= 2u) {
          uint4 *pz = reinterpret_cast<uint4 *>(blk);
#pragma unroll
          for (int i = 0; i < RV; ++i) pz[i] = make_uint4(0u, 0u, 0u, 0u);
        }
      }
      WAVE_SYNC();
      GG_PROF(3);

      // phase 3 - all sixteen boards in ONE pass, RPL adjacent rows per lane: patch the classes, resolve captures and
      // ko, the next mover's mask.  Conditions are kept as 0 / ~0 words in VGPRs and applied with bit operations: a
      // boolean that goes through v_cmp -> s_and_b64 -> v_cndmask (or through an EXEC branch around three instructions)
      // is a VALU -> SALU -> VALU round trip on this wave's critical path, and with 16 boards per wave the "rare"
      // capture path runs on 85 % of the plies (some board of the wave captures), so it is straight-line code too.
      {
        const int a = a_q;
        const uint32_t fl = fl_q;
        const uint4 cq = *reinterpret_cast<const uint4 *>(clsv + 4 * s4);
        const uint32_t c0 = cq.x, c1 = cq.y, c2 = cq.z, c3 = cq.w;
        const int turn0 = fl & 1u;
        uint32_t *pmine = st + turn0 * PL + s4 * RS + r0;
        uint32_t *popp = st + (1 - turn0) * PL + s4 * RS + r0;
        const uint32_t *gr = sc + s4 * SCB + r0;   // block j: gr[j * RS + r], j = 4: the board's G block (valid when q has a friendly neighbour)
        uint32_t mine1[RPL], opp0[RPL], b0[RPL], b1[RPL], b2[RPL], b3[RPL], bg[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          mine1[r] = pmine[r];   // (rows >= N are zero)
          opp0[r] = popp[r];
          b0[r] = gr[r]; b1[r] = gr[RS + r]; b2[r] = gr[2 * RS + r]; b3[r] = gr[3 * RS + r];
          bg[r] = gr[4 * RS + r];
        }
        const bool moves_now = a >= 0;
        const bool is_pass = a == hf.P;
        const uint32_t stone_m = (moves_now && !is_pass) ? ~0u : 0u;
        int ar, ac;
        split_action(a, N, hf.inv, ar, ac);                       // (garbage for a pass / an idle board: masked below)
        // captured = an opponent group with no liberty left (CL_CAPT, set by its flood lane; never on a board that does not move)
        const uint32_t km0 = (uint32_t)__builtin_amdgcn_sbfe((int)c0, 5, 1), km1 = (uint32_t)__builtin_amdgcn_sbfe((int)c1, 5, 1),
                       km2 = (uint32_t)__builtin_amdgcn_sbfe((int)c2, 5, 1), km3 = (uint32_t)__builtin_amdgcn_sbfe((int)c3, 5, 1);
        const uint32_t capt_m = km0 | km1 | km2 | km3;
        uint32_t g0[RPL], gch[RPL], cap[RPL], Mm_fix[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          // (no row mask: a flood block's rows >= N are zero - the flooded colour has no stone there - and so is word R)
          g0[r] = bg[r] & stone_m;   // the G block: the flood of G, or the stone alone as phase 1 left it there
          gch[r] = B3(b0[r], b1[r], b2[r], T_OR3) | b3[r];   // the opponent groups whose class changes (a lane that flooded G or nothing left its block empty)
          cap[r] = B3(b3[r], km3, B3(b2[r], km2, B3(b1[r], km1, b0[r] & km0, T_ANDOR), T_ANDOR), T_ANDOR);
          Mm_fix[r] = 0u;
        }
        uint32_t libsG = (B3(c0, c1, c2, T_OR3) | c3) >> 6 & 3u;   // liberties of G among the empty points (saturated at 2)
        uint32_t kor[RPL];   // the ko point as rows of this lane (almost always none)
#pragma unroll
        for (int r = 0; r < RPL; ++r) kor[r] = 0u;
        if (__ballot(capt_m != 0u)) {   // a capture on some board of the wave (85 % of the plies at 16 boards per wave)
          // Captured stones next to G are liberties of G too.  Every captured group holds a neighbour of q, and q is part
          // of G: with ncapn captured neighbours G gains at least ncapn liberties.  Only when that leaves the count below
          // two (no empty liberty, one captured neighbour) do the captured points next to G have to be counted (rare)
          const uint32_t ncapn = 0u - (km0 + km1 + km2 + km3);    // masks are 0 / -1
          if (__ballot(ncapn == 1u && libsG == 0u)) {
            uint32_t dg[RPL];
            dilate_rows<RPL>(g0, dg);
            uint32_t cntc = 0;
#pragma unroll
            for (int r = 0; r < RPL; ++r) cntc += (uint32_t)__popc(dg[r] & cap[r]);
            const uint32_t tot = quad_sum(cntc < 2u ? cntc : 2u);
            libsG += (ncapn == 1u && libsG == 0u) ? (tot < 2u ? tot : 2u) : ncapn;
          } else {
            libsG += ncapn;
          }
          // gogame.py:72-75: ko iff exactly one stone died and the new stone is boxed in
          const uint32_t ncap1 = ((c0 >> 2) & km0 & 1u) + ((c1 >> 2) & km1 & 1u) + ((c2 >> 2) & km2 & 1u) + ((c3 >> 2) & km3 & 1u);
          const bool ko = !(c0 & CL_OPEN) && ncapn == 1u && ncap1 == 1u;
          // the one captured stone is q's neighbour in the direction of its flood lane (0 up, 1 down, 2 left, 3 right; masks are 0 / -1)
          const uint32_t kr = (uint32_t)ar + km0 - km1 - (uint32_t)r0;
          const uint32_t ko_oh = (ko && kr < (uint32_t)RPL) ? (1u << (kr & 31)) : 0u;
          if (__ballot(ko_oh != 0u)) {
            const uint32_t ko_bit = 1u << (((uint32_t)ac + km2 - km3) & 31u);
#pragma unroll
            for (int r = 0; r < RPL; ++r) kor[r] = (uint32_t)__builtin_amdgcn_sbfe((int)ko_oh, r, 1) & ko_bit;
          }
          // the mover's groups in atari next to a captured stone (and not merged into G) now have >= 2 liberties.  (Most
          // capturing boards have no such group at all: the neighbourhood of the captured stones is only worked out when
          // some capturing board of the wave has a mover's stone in atari outside G)
          uint32_t atari[RPL], f[RPL];
          uint32_t anya = 0;
#pragma unroll
          for (int r = 0; r < RPL; ++r) { atari[r] = B3(mine1[r], M[r], g0[r], TA & ~(TB | TC) & 0xFF); f[r] = 0u; anya |= atari[r]; }
          uint32_t anyf = 0;
          if (__ballot(anya != 0u && capt_m != 0u)) {
            dilate_rows<RPL>(cap, f);
#pragma unroll
            for (int r = 0; r < RPL; ++r) { f[r] &= atari[r]; anyf |= f[r]; }
          }
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
            for (int r = 0; r < RPL; ++r) Mm_fix[r] = f[r];
          }
        }
        const uint32_t gsel = libsG >= 2u ? ~0u : 0u;
        uint32_t Mo2[RPL], opp1[RPL], Mm2[RPL], e[RPL], x[RPL], nbr[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          Mo2[r] = B3(M[r], opp0[r], gch[r], TA & TB & ~TC & 0xFF);         // (a captured group was in atari: never in M)
          opp1[r] = opp0[r] & ~cap[r];
          const uint32_t Mm = B3(M[r], mine1[r], g0[r], TA & TB & ~TC & 0xFF);
          Mm2[r] = B3(gsel, g0[r], Mm, T_ANDOR) | Mm_fix[r];               // (M & mine & ~g0) | (gsel & g0)
          // state_utils.compute_invalid_moves on the lane's rows (invalid_from2, RPL rows per lane)
          e[r] = B3(full[r], opp1[r], mine1[r], TA & ~(TB | TC) & 0xFF);
          x[r] = B3(mine1[r], Mm2[r], e[r], (TA & ~TB & 0xFF) | TC) | Mo2[r];
        }
        dilate_rows<RPL>(x, nbr);
        const uint32_t mv_m = moves_now ? ~0u : 0u;
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          const uint32_t invalid = B3(e[r], nbr[r], full[r], ~(TA & TB) & TC & 0xFF) | kor[r];
          inv_r[r] = B3(mv_m, invalid, inv_r[r], T_SEL);
          // (a board that does not move, or passes, has no flood, no G and no capture: the classes below are M again)
          M[r] = Mm2[r] | Mo2[r];
        }
        if (capt_m) {
#pragma unroll
          for (int r = 0; r < RPL; ++r) popp[r] = opp1[r];
        }
        if (moves_now && t5 == 0) {
          const uint32_t passed0 = (fl >> 1) & 1u, done0 = (fl >> 2) & 1u;
          const uint32_t passed = is_pass ? 1u : 0u, done = done0 | (passed & passed0);
          flagsv[s4] = (uint32_t)(turn0 ^ 1) | (passed << 1) | (done << 2) | 8u;
          lastv[s4] = a;
          playedv[s4] += 1;
        }
      }
      WAVE_SYNC();
      GG_PROF(4);
    }
    if (plies >= 8) fair.release();
    GG_PROF(5);   // (nothing between the last ply and the write-back)

    // ---------------------------------------------------------------- store
    // The lane-derived values of the write-back are recomputed from a fresh (volatile) lane id: hoisted above the ply
    // loop they are spilled there and every reload is a scratch round trip on the tail of a short launch.
    int lnS;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lnS));
    const Half hs = make_half(lnS, N, inv);
    const int q4s = lnS >> 2, t4s = lnS & 3, r04s = RPL * t4s;
    const bool rowS = hs.hl < RS;
    WAVE_SYNC();
    if (CACHED && env.canonical) {
      // canonical_form (gogame.py:313-321) of a board whose next mover is white: colours swapped, turn plane cleared -
      // done on the LDS planes, so that the byte planes and the workspace rows written below agree
      const uint32_t fc = flagsv[q4s];
      if ((fc & 1u) && playedv[q4s] != 0) {
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          const uint32_t bk = st[0 * PL + q4s * RS + r04s + r], wh = st[1 * PL + q4s * RS + r04s + r];
          st[0 * PL + q4s * RS + r04s + r] = wh;
          st[1 * PL + q4s * RS + r04s + r] = bk;
        }
      }
      WAVE_SYNC();
      if (hs.lane < kNB4 && (flagsv[hs.lane] & 1u) && playedv[hs.lane] != 0) flagsv[hs.lane] &= ~1u;
      WAVE_SYNC();
    }
    if (TRACKED || CACHED) {
      // park the register rows, then one flat coalesced copy of the group's contiguous block
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        if (r04s + r < RS) {
          const uint32_t bk = st[0 * PL + q4s * RS + r04s + r], wh = st[1 * PL + q4s * RS + r04s + r];
          park[0 * PL + q4s * RS + r04s + r] = inv_r[r];
          park[1 * PL + q4s * RS + r04s + r] = M[r] & bk;
          park[2 * PL + q4s * RS + r04s + r] = M[r] & wh;
        }
      }
      WAVE_SYNC();
      const int64_t nbrd = (B - b_first) < nb ? (B - b_first) : nb;
      const int nw = (int)nbrd * W;
      uint32_t *gp = (CACHED ? env.ws : reinterpret_cast<uint32_t *>(states)) + b_first * (int64_t)W;
      const uint32_t invW = ((1u << 20) + (uint32_t)W - 1u) / (uint32_t)W;
      // untouched boards are not rewritten: one bit per board, read once (the flag words sat in every round of the loop, in
      // front of the row read: two dependent LDS round trips per 64 words); four words per lane and round, their LDS reads in
      // flight together (round 6: the env step of 65 536 games 39.4 -> 37.4 us, a one-ply tracked launch 24.6 -> 23.5)
      bool tch = false;
      if (hs.lane < (int)nbrd) tch = playedv[hs.lane] != 0 || (flagsv[hs.lane] & 32u);
      const uint64_t tmask = __ballot(tch);
#pragma unroll 1
      for (int i0 = hs.lane; i0 < nw; i0 += 4 * kWave) {
        uint32_t v[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k * kWave;
          const int ic = i < nw ? i : 0;
          const int sb = (int)(((uint32_t)ic * invW) >> 20), w = ic - sb * W;
          ok[k] = i < nw && ((tmask >> sb) & 1ull);
          const int pl = (int)(((uint32_t)w * hs.inv) >> 16), rw = w - pl * N;   // (w == 5 N: pl == 5, rw == 0)
          const uint32_t *src = pl >= 5 ? flagsv + sb : (pl < 2 ? st + pl * PL + sb * RS + rw : park + (pl - 2) * PL + sb * RS + rw);
          const uint32_t x = *src;
          v[k] = pl >= 5 ? (x & 7u) : x;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (ok[k]) gp[i0 + k * kWave] = v[k];
      }
      if (hs.lane < nb && b_first + hs.lane < B) {
        const int sb = hs.lane;
        const int64_t b = b_first + sb;
        const int played = playedv[sb];
        if (CACHED) {
          if (env.status) env.status[b] = played ? GG_STATUS_OK : GG_STATUS_ILLEGAL;
        } else {
          // (drawn moves: the generator the launch found + one step per ply played - see the draw; policy-weighted moves: as the draw block left it)
          if (!MOVES || WTS) rng[b] = (((uint64_t)rngv[2 * sb + 1] << 32) | rngv[2 * sb]) + (MOVES ? 0ull : (uint64_t)(uint32_t)played * 0x9E3779B97F4A7C15ull);
          if (last_actions) last_actions[b] = lastv[sb];
          if (steps_done && played) atomicAdd(reinterpret_cast<unsigned long long *>(steps_done) + b, (unsigned long long)played);   // (no read-back: nothing to wait for)
          if (MOVES && played_out) played_out[b] = played;
        }
      }
      WAVE_SYNC();
    }
    if (CACHED) {
      // the output batch: the new position of every game of the group in one contiguous write; a refused move (rare)
      // gets the input row instead, copied afterwards (stores of one wave to one address land in program order)
      const int nbrd = (int)((B - b_first) < nb ? (B - b_first) : nb);
      emit_group<R, RPL>(env.states_out + b_first * (int64_t)S, nbrd, N, st, PL, RS, inv_r, flagsv,
                         lds + Lds4<R>::kGrpBits, reinterpret_cast<uint2 *>(lds + Lds4<R>::kGrpLut), hs.lane);
      bool refused_any = false;
      if (hs.lane < nbrd) refused_any = playedv[hs.lane] == 0;
      if (__ballot(refused_any)) {
#pragma unroll 1
        for (int i = 0; i < nb / 2; ++i) {
          const int s = 2 * i + hs.h;
          if (s < nbrd && playedv[s] == 0) {
            const int64_t b = b_first + s;
            copy_row_h(states + b * (int64_t)S, env.states_out + b * (int64_t)S, S, hs.hl, true);
          }
        }
      }
    }
    if (ENV) {
      // ---- GoEnv.step outputs.  Tromp-Taylor areas (gym_go/gogame.py:275-300, areas16): needed for a finished game
      // (reward `real`) or for every game (`heuristic`).
      const uint32_t fl = flagsv[q4s];
      const bool onb = (fl >> 3) & 1u;
      const bool doneb = (fl >> 2) & 1u;
      int area_b = 0, area_w = 0;
      if (__ballot(onb && (env.heuristic != 0 || doneb))) {
        // two floods of the empty points per board, in lanes 2 s + c of the lower half (the flood blocks of the plies are
        // free by now); the counts reach the boards' quads through the class words
        const uint32_t cnt = areas16<R, FULLN>(st, sc, sc + 32 * RS, N, lnS);
        WAVE_SYNC();
        if (lnS < 32) clsv[lnS] = cnt;
        WAVE_SYNC();
        area_b = (int)clsv[2 * q4s];
        area_w = (int)clsv[2 * q4s + 1];
      }
      if (t4s == 0 && onb) {
        const int64_t b = b_first + q4s;
        const float margin = (float)(area_b - area_w) - env.komi;
        float rwd;   // GoEnv.reward (gym_go/envs/go_env.py:128-149), black's perspective
        if (env.heuristic) rwd = doneb ? (margin > 0.f ? 1.f : -1.f) * (float)hs.P : margin;
        else rwd = doneb ? (margin > 0.f ? 1.f : (margin < 0.f ? -1.f : 0.f)) : 0.f;
        if (env.rewards) env.rewards[b] = rwd;
        if (env.dones) env.dones[b] = (uint8_t)doneb;
        if (env.status) env.status[b] = ((fl >> 4) & 1u) ? GG_STATUS_ILLEGAL : GG_STATUS_OK;
        if (env.taken) env.taken[b] = MOVES ? (WTS ? wact[q4s] : mv_first) : lastv[q4s];   // (one ply per launch: the move of the ply)
      }
      if (env.states_out) {
        // the observation: every board of the group as byte planes, one contiguous write
        const int nbrd = (int)((B - b_first) < nb ? (B - b_first) : nb);
        emit_group<R, RPL>(env.states_out + b_first * (int64_t)S, nbrd, N, st, PL, RS, inv_r, flagsv,
                           lds + Lds4<R>::kGrpBits, reinterpret_cast<uint2 *>(lds + Lds4<R>::kGrpLut), hs.lane);
      }
    }
    if (IO == 0) {
      // byte planes in place: the whole group in one contiguous write (a board that did not move is rewritten with what
      // was loaded from it), then the per-game outputs
      const int nbrd = (int)((B - b_first) < nb ? (B - b_first) : nb);
      bool any_wr = false;
      if (hs.lane < nbrd) any_wr = playedv[hs.lane] != 0 || (flagsv[hs.lane] & 32u);
      if (__ballot(any_wr))
        emit_group<R, RPL>(states + b_first * (int64_t)S, nbrd, N, st, PL, RS, inv_r, flagsv, lds + Lds4<R>::kGrpBits,
                           reinterpret_cast<uint2 *>(lds + Lds4<R>::kGrpLut), hs.lane);
      if (hs.lane < nbrd) {
        const int sb = hs.lane;
        const int64_t b = b_first + sb;
        const int played = playedv[sb];
        if (!MOVES) rng[b] = (((uint64_t)rngv[2 * sb + 1] << 32) | rngv[2 * sb]) + (uint64_t)(uint32_t)played * 0x9E3779B97F4A7C15ull;
        if (last_actions) last_actions[b] = lastv[sb];
        if (steps_done && played) atomicAdd(reinterpret_cast<unsigned long long *>(steps_done) + b, (unsigned long long)played);
        if (MOVES && played_out) played_out[b] = played;
      }
      WAVE_SYNC();
    }
#pragma unroll 1
    for (int i = 0; i < (PACKED ? nb / 2 : 0); ++i) {
      if ((hs.lane >> 3) == i) {   // the pair's owner quads hand their mask rows over
        uint32_t *tp = tmp + ((q4s & 1) * 2) * RS + r04s;
#pragma unroll
        for (int r = 0; r < RPL; ++r) tp[r] = inv_r[r];
      }
      WAVE_SYNC();
      const int s = 2 * i + hs.h;
      const uint32_t fl = flagsv[s];
      const bool on = (fl >> 3) & 1u;
      const int64_t b = on ? b_first + s : B - 1;
      const int played = playedv[s];
      uint32_t black = 0, white = 0, invalid = 0;
      if (rowS) {
        black = st[0 * PL + s * RS + hs.hl];
        white = st[1 * PL + s * RS + hs.hl];
        invalid = tmp[(hs.h * 2) * RS + hs.hl];
      }
      const bool wr = on && (played != 0 || (fl & 32u));
      if (PACKED) {
        store_packed_h(reinterpret_cast<uint32_t *>(states) + b * (int64_t)W, N, hs, black, white, invalid, fl & 1u,
                       (fl >> 1) & 1u, (fl >> 2) & 1u, wr);
      } else if (__ballot(wr)) {
        emit_store_h<R>(states + b * (int64_t)S, black, white, invalid, fl & 1u, (fl >> 1) & 1u, (fl >> 2) & 1u, hs,
                        v2 + hs.h * 128, lut, wr);
      }
      if (on && hs.hl == 0) {
        if (!MOVES) rng[b] = (((uint64_t)rngv[2 * s + 1] << 32) | rngv[2 * s]) + (uint64_t)(uint32_t)played * 0x9E3779B97F4A7C15ull;
        if (last_actions) last_actions[b] = lastv[s];
        if (steps_done && played) atomicAdd(reinterpret_cast<unsigned long long *>(steps_done) + b, (unsigned long long)played);
        if (MOVES && played_out) played_out[b] = played;
      }
      WAVE_SYNC();
    }
    GG_PROF(7);   // write-back
    GG_PROF_FLUSH;
  }
  GG_WHERE_END;
}

// byte planes -> tracked boards: the rows of planes 0 / 1 / 3 and the liberty classes of one v2 analysis
template <int R>
__global__ __launch_bounds__(kWave, 4) void k_track(const uint8_t *__restrict__ states, uint32_t *__restrict__ tracked,
                                                     int64_t B, int N, uint32_t inv, AgeSplit age) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[Lds2<R>::kTotal];
  const Half hf = make_half(threadIdx.x, N, inv);
  bool tables = false;
  const int S = 6 * hf.P, W = 5 * N + 1;
  uint8_t *io = reinterpret_cast<uint8_t *>(lds) + hf.h * Cfg<R>::kIoBytes;
  const int64_t npairs = (B + 1) >> 1;
  const PairSpan span = pair_span(npairs, age);
  for (int64_t p = span.first; p < span.end; p += span.stride) {
    const bool on = 2 * p + hf.h < B;
    const int64_t b = on ? 2 * p + hf.h : B - 1;
    const uint8_t *gs = states + b * (int64_t)S;
    PairRegs<R> pr;
    pair_issue<R>(pr, gs, 4 * hf.P, hf, tables);
    uint32_t flags;
    const uint32_t mi = pair_commit<R>(pr, gs, 4 * hf.P, io, hf, lds, nullptr, tables, flags);
    const uint32_t black = plane_to_row<R>(io + mi, N, hf.hl);
    const uint32_t white = plane_to_row<R>(io + mi + hf.P, N, hf.hl);
    const uint32_t invalid = plane_to_row<R>(io + mi + 3 * hf.P, N, hf.hl);
    uint32_t mb, ab, mw;
    analyze2<R, false>(black, white, hf.full_l1 & ~(black | white), hf, lds, mb, ab, mw);
    uint32_t *gp = tracked + b * (int64_t)W;
    if (on && hf.hl < N) {
      gp[hf.hl] = black; gp[N + hf.hl] = white; gp[2 * N + hf.hl] = invalid;
      gp[3 * N + hf.hl] = mb; gp[4 * N + hf.hl] = mw;
    }
    if (on && hf.hl == 31) gp[5 * N] = (flags & 1u) | ((flags >> 1) & 6u);
  }
}

}  // namespace gg
