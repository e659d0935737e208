// gg_v3.h - fused uniform-random rollout, TWELVE BOARDS PER WAVEFRONT, incremental liberty classes.
#pragma once
#include "gg_v2.h"

namespace gg {

// ===================================================================== v3: incremental analysis, 12 boards per wave
// The v2 rollout re-derives every group's liberty class each ply with 22 floods per board (2 boards per wave).  The cost
// of a flood batch does not depend on how many of the wave's 64 lanes carry a flood, so the lever is floods per board:
// a move at q only changes the groups ADJACENT to q (and, rarely, to a captured group).  The kernel therefore keeps, per
// board, the stones of either colour whose group has >= 2 liberties (every other stone is in atari: a legal position
// has no liberty-less group) and updates them per ply from FIVE floods:
//   lane 0 of a board: the mover's group G that the new stone joins (flood from q through the mover's stones + q);
//   lanes 1-4:         the opponent's group at the upper / lower / left / right neighbour of q (empty otherwise).
// A ply is three phases, all on ONE lane assignment (board = lane / 5; lanes 60-63 idle):
//   1. sampling, five lanes per board and four rows each: liveness, the generator (drawn redundantly by the five lanes),
//      the k-th valid point of the stored mask (a lane counts its own rows, the board's prefix / total come from wave
//      shifts, the lane that holds the point selects row and bit), the stone ORed into the mover's plane; auto-reset on
//      a rare path (one lane per board and 19 rows per lane: 5.25e9 steps/s against 5.5e9);
//   2. one lane per (board, role): the flood, then the liberties (dilate & empty, saturated at 2) and the size of the
//      lane's own group, 19 rows in registers; an opponent group that keeps >= 2 liberties zeroes its result;
//   3. all twelve boards in one pass, four adjacent rows per lane (lane -> board lane / 5 as in phase 2, rows 4t .. 4t+3
//      with t = lane % 5).  One row per lane, three boards per pass and four passes ran 4.7e9 steps/s, two rows / six
//      boards / two passes 5.1e9, this 5.25e9: the work of a pass that does not depend on the row is paid per pass:
//        * an opponent group next to q with no liberty left is captured, with one left it leaves the class plane;
//        * G takes the class of its own count (+ the captured points next to it);
//        * a mover's group in atari next to a captured stone gains a liberty -> multi (rare; a flood through the atari
//          set in this layout);
//        * every other group keeps its class.  The invalid-move mask follows from the classes exactly as in v2.
// 12 boards x 5 lanes = 60 lanes per flood batch: the floods cost a sixth per board - about 100 VALU ops per board and
// ply against 346 (PMC, profiles/r01_summary.md).
// Board state lives in LDS between the phases (5 rows per board: black, white, invalid, multi_black, multi_white);
// 10 224 B per wave, 128 VGPRs: four waves per SIMD.
constexpr int kNB3 = 12;

template <int R>
struct Lds3 {
  static constexpr int RS = Cfg<R>::kRowStride;
  static constexpr int kState = 0;                               // [5][kNB3][RS]
  static constexpr int kMeta = kState + 5 * kNB3 * RS;           // flags[12], act[12], last[12], played[12], rng[24]
  static constexpr int kUnion = kMeta + 72;
  // ply loop: per flood lane its result word (liberty class, size, ...) + the transpose buffer of the group masks
  // (60 flood lanes + one shared dummy block for the four idle lanes, whose floods are empty)
  static constexpr int kCls = kUnion;
  static constexpr int kSc = kCls + kWave;
  static constexpr int kLoopEnd = kSc + (5 * kNB3 + 1) * RS;
  // load / store: the v2 analysis in its compact form (region 0 only: staging / transpose buffer); at store time the
  // emitter's scratch (2 x 128 words) and the spread table (uint2[256])
  static constexpr int kV2 = kUnion;
  static constexpr int kLut = kV2 + 256;
  static constexpr int kIoEnd = kV2 + (Lds2<R>::kRegion0 > 768 ? Lds2<R>::kRegion0 : 768);
  static constexpr int kTotal = kLoopEnd > kIoEnd ? kLoopEnd : kIoEnd;
};

// is the point set x (L1 rows of this half) non-empty / larger than one point?
__device__ __forceinline__ void set_size(uint32_t x, const Half &hf, bool &any, bool &two) {
  const uint32_t nz = half_of(__ballot(x != 0), hf.h), many = half_of(__ballot(__popc(x) > 1), hf.h);
  any = nz != 0;
  two = ((nz & (nz - 1u)) | many) != 0;
}
// phase 3 keeps FOUR adjacent rows (4t .. 4t+3) per lane, five lanes per board, all twelve boards in one pass - the lane
// assignment of the floods (board = lane / 5).  The 5 ballot bits of board k:
__device__ __forceinline__ uint32_t fifth_of(uint64_t ballot, int k) { return (uint32_t)(ballot >> (5 * k)) & 0x1Fu; }
// 4-neighbourhood dilation of the four rows of a lane: the row above x[0] / below x[3] sits in the neighbouring lane (a
// non-existent row >= R is zero, so nothing leaks from the previous board; what leaks into such a row from the next
// board is masked by the caller).  The centre point is not part of the result.
__device__ __forceinline__ void dilate_quad(const uint32_t (&x)[4], uint32_t (&d)[4]) {
  const uint32_t up = dpp0<0x138>(x[3]), dn = dpp0<0x130>(x[0]);
  d[0] = B3(shl1(x[0]), x[0] >> 1, up, T_OR3) | x[1];
  d[1] = B3(shl1(x[1]), x[1] >> 1, x[0], T_OR3) | x[2];
  d[2] = B3(shl1(x[2]), x[2] >> 1, x[1], T_OR3) | x[3];
  d[3] = B3(shl1(x[3]), x[3] >> 1, x[2], T_OR3) | dn;
}
__device__ __forceinline__ void unpack4(const uint4 v, uint32_t (&x)[4]) { x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }

// MOVES: the moves are given (moves: int32 [B][plies], gg_batch_play_moves) instead of drawn: a game stops at its first
// move that is out of range, on an invalid point or made after the game has ended; played_out[b] = moves applied.
// IO: 0 = byte planes (uint8 [B][6][N][N]), 1 = packed boards (uint32 [B][3N+1]), 2 = TRACKED boards (uint32 [B][5N+1]:
// the rows of black, white, invalid, multi_black, multi_white + the flag word - a packed board that carries its
// liberty classes, so that a launch needs no first analysis: per-ply stepping at the fused kernel's rate).
// FULLN: the board fills the row capacity (N == R: 9, 13, 19) - the per-row "r < N" guards of the lane-per-board code
// fold away at compile time (they cost one v_cndmask each otherwise).
template <int R, int IO, bool MOVES = false, bool FULLN = false>
__global__ __launch_bounds__(kWave, 4) void k_rollout3(uint8_t *__restrict__ states, uint64_t *__restrict__ rng,
                                                       int32_t *__restrict__ last_actions, int64_t *__restrict__ steps_done,
                                                       int64_t B, int N, uint32_t inv, int plies, int auto_reset,
                                                       int nb, const int32_t *__restrict__ moves = nullptr,
                                                       int32_t *__restrict__ played_out = nullptr) {
  constexpr int RS = Lds3<R>::RS;
  constexpr int RV = (R + 3) / 4;
  constexpr int PL = kNB3 * RS;   // words per plane of all boards
  __shared__ __attribute__((aligned(16))) uint32_t lds[Lds3<R>::kTotal];
  if (FULLN) N = R;   // a compile-time constant from here on: row masks, r * N + c and the "row exists" tests fold
  const Half hf = make_half(threadIdx.x, N, inv);
  uint32_t *st = lds + Lds3<R>::kState;     // st[p * PL + s * RS + row]
  uint32_t *flagsv = lds + Lds3<R>::kMeta;  // bit 0 turn, 1 passed, 2 done, 3 on
  int *actv = reinterpret_cast<int *>(lds + Lds3<R>::kMeta + 12);
  int *lastv = reinterpret_cast<int *>(lds + Lds3<R>::kMeta + 24);
  int *playedv = reinterpret_cast<int *>(lds + Lds3<R>::kMeta + 36);
  uint32_t *rngv = lds + Lds3<R>::kMeta + 48;   // [2 * s], [2 * s + 1]
  uint32_t *clsv = lds + Lds3<R>::kCls;
  uint32_t *sc = lds + Lds3<R>::kSc;
  uint32_t *v2 = lds + Lds3<R>::kV2;
  uint2 *lut = reinterpret_cast<uint2 *>(lds + Lds3<R>::kLut);
  constexpr bool PACKED = IO == 1, TRACKED = IO == 2;
  const int S = 6 * hf.P, W = (TRACKED ? 5 : 3) * N + 1;
  const bool row = hf.hl < RS;
  // nb (even, <= kNB3) boards per wave: the host picks it so that the groups fill the resident waves evenly
  const int64_t ngroups = (B + nb - 1) / nb;

  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int64_t b_first = g * nb;
    // ---------------------------------------------------------------- load
    WAVE_SYNC();
    if (TRACKED) {
      // the group's boards are ONE contiguous block of nb x (5 N + 1) words: a flat, fully coalesced copy into the LDS
      // planes (all loads in flight at once), instead of six dependent per-pair passes
      const int64_t nbrd = (B - b_first) < nb ? (B - b_first) : nb;
      const int nw = (int)nbrd * W;
      const uint32_t *gp = reinterpret_cast<const uint32_t *>(states) + b_first * (int64_t)W;
      for (int i = hf.lane; i < 5 * PL; i += kWave) st[i] = 0;      // rows N .. RS-1 and absent boards read as zero
      const uint32_t invW = ((1u << 20) + (uint32_t)W - 1u) / (uint32_t)W;   // i / W exactly for i < 12 * 96
      WAVE_SYNC();
      for (int i = hf.lane; i < nw; i += kWave) {
        const uint32_t v = gp[i];
        const int sb = (int)(((uint32_t)i * invW) >> 20), w = i - sb * W;
        if (w == 5 * N) {
          flagsv[sb] = (v & 7u) | 8u;
        } else {
          const int pl = (int)(((uint32_t)w * hf.inv) >> 16), rw = w - pl * N;
          st[pl * PL + sb * RS + rw] = v;
        }
      }
      if (hf.lane < nb) {
        const int sb = hf.lane;
        const bool on = b_first + sb < B;
        if (!on) flagsv[sb] = 0;
        lastv[sb] = -1;
        playedv[sb] = 0;
        if (!MOVES) {
          const uint64_t x = rng[on ? b_first + sb : B - 1];
          rngv[2 * sb] = (uint32_t)x;
          rngv[2 * sb + 1] = (uint32_t)(x >> 32);
        }
      }
      WAVE_SYNC();
    }
#pragma unroll 1
    for (int i = 0; i < (TRACKED ? 0 : nb / 2); ++i) {   // byte planes / packed boards: 6 pairs, first classes by the v2 analysis
      const int s = 2 * i + hf.h;
      const bool on = b_first + s < B;
      const int64_t b = on ? b_first + s : B - 1;
      uint32_t black, white, invalid, mb = 0, mw = 0;
      int turn, passed, done;
      if (TRACKED) {
        const uint32_t *gp = reinterpret_cast<const uint32_t *>(states) + b * (int64_t)W;
        black = white = invalid = 0;
        if (hf.hl < N) {
          black = gp[hf.hl]; white = gp[N + hf.hl]; invalid = gp[2 * N + hf.hl];
          mb = gp[3 * N + hf.hl]; mw = gp[4 * N + hf.hl];
        }
        const uint32_t fw = gp[5 * N];
        turn = fw & 1u; passed = (fw >> 1) & 1u; done = (fw >> 2) & 1u;
      } else if (PACKED) {
        uint32_t fw;
        load_packed_h(reinterpret_cast<const uint32_t *>(states) + b * (int64_t)W, N, hf, black, white, invalid, fw);
        turn = fw & 1u; passed = (fw >> 1) & 1u; done = (fw >> 2) & 1u;
      } else {
        const uint8_t *gs = states + b * (int64_t)S;
        uint8_t *io = reinterpret_cast<uint8_t *>(v2) + hf.h * Cfg<R>::kIoBytes;
        const uint32_t flags = load_flags_h(gs, hf.P, 0, hf);
        WAVE_SYNC();
        const uint32_t mi = stage_in_h(gs, 4 * hf.P, io, hf.hl);
        WAVE_SYNC();
        black = plane_to_row<R>(io + mi, N, hf.hl);
        white = plane_to_row<R>(io + mi + hf.P, N, hf.hl);
        invalid = plane_to_row<R>(io + mi + 3 * hf.P, N, hf.hl);
        turn = flags & 1u; passed = (flags >> 2) & 1u; done = (flags >> 3) & 1u;
      }
      if (!TRACKED) {
        uint32_t ab;
        analyze2<R, false>(black, white, hf.full_l1 & ~(black | white), hf, v2, mb, ab, mw, nullptr, nullptr, true);
      }
      if (row) {
        st[0 * PL + s * RS + hf.hl] = black;
        st[1 * PL + s * RS + hf.hl] = white;
        st[2 * PL + s * RS + hf.hl] = invalid;
        st[3 * PL + s * RS + hf.hl] = mb;
        st[4 * PL + s * RS + hf.hl] = mw;
      }
      if (hf.hl == 0) {
        flagsv[s] = (uint32_t)turn | ((uint32_t)passed << 1) | ((uint32_t)done << 2) | (on ? 8u : 0u);
        lastv[s] = -1;
        playedv[s] = 0;
        if (!MOVES) {
          const uint64_t x = rng[b];
          rngv[2 * s] = (uint32_t)x;
          rngv[2 * s + 1] = (uint32_t)(x >> 32);
        }
      }
      WAVE_SYNC();
    }

    // ---------------------------------------------------------------- the plies
    int mv_next = 0;
#pragma unroll 1
    for (int t = 0; t < plies; ++t) {
      // the lane-derived indices of the three phases are recomputed every ply (a few VALU ops) instead of being hoisted
      // out of the loop, where they end up in scratch: a reload is a vector-memory round trip at the top of each phase
      // (volatile asm: neither hoisted nor merged)
      int ln;   // = hf.lane (one wave per workgroup), straight from the hardware
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      const int s5 = (ln * 13) >> 6, t5 = ln - 5 * s5;   // phases 2 and 3: board and role / row quad of this lane
      __builtin_assume(t5 >= 0 && t5 < 5);
      const int s5c = s5 < kNB3 ? s5 : 0, r0 = 4 * t5;
      uint32_t full4[4];   // the N-bit row mask of the lane's rows that exist (also what keeps the unwritten rows >= R of
                           // the flood blocks out: flood results only have bits < N)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        full4[r] = (s5 < kNB3 && r0 + r < N) ? (1u << N) - 1u : 0u;
      }
      // phase 1 - five lanes per board, four rows each (the lane assignment of phases 2 and 3): liveness, the generator
      // (drawn redundantly by the five lanes), the k-th valid point of the stored mask (or the given move).  With one lane
      // per board the 19 row counts and the row selection cost 19 steps on 12 useful lanes; here a lane counts its own
      // four rows, the board's prefix / total come from wave shifts, and only the lane that holds the point selects.
      uint64_t resetm;
      {
        const bool bl = s5 < nb;
        const int sb = s5c;
        const uint32_t fl = flagsv[sb];
        const bool on = bl && ((fl >> 3) & 1u);
        const bool done = (fl >> 2) & 1u;
        bool live, reset, place = false, wr_act;
        int rabs = 0, a;
        uint32_t pos = 0;
        uint64_t x = 0;
        if (MOVES) {
          // the move of this ply was fetched during the previous one (mv_next), the next one is requested now
          const int64_t bm = (b_first + sb < B) ? b_first + sb : B - 1;
          const int mv = t == 0 ? moves[bm * (int64_t)plies] : mv_next;
          if (t + 1 < plies) mv_next = moves[bm * (int64_t)plies + t + 1];
          reset = false;
          live = on && !done && !((fl >> 4) & 1u) && mv >= 0 && mv <= hf.P;
          a = hf.P;
          if (live && mv < hf.P) {
            int ar, ac;
            split_action(mv, N, hf.inv, ar, ac);
            live = ((st[2 * PL + sb * RS + ar] >> ac) & 1u) == 0;
            rabs = ar; pos = (uint32_t)ac;
            a = mv;
          }
          wr_act = bl && t5 == 0;
          if (on && !live && wr_act) flagsv[sb] = fl | 16u;   // stopped for good
          if (!live) a = -1;
          place = wr_act && a >= 0 && a < hf.P;
        } else {
          live = on && !(done && !auto_reset);
          reset = live && done;           // auto-reset: the board is init_state from now on
          uint32_t v[4];
          {
            uint32_t iv[4];
            unpack4(*reinterpret_cast<const uint4 *>(st + 2 * PL + sb * RS + (r0 < R ? r0 : 0)), iv);
            const uint32_t rm = reset ? ~0u : 0u;   // a board being reset plays on the empty board
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = B3(full4[r], rm, iv[r], TA & (TB | (~TC & 0xFF)));   // full4: 0 for rows >= N
          }
          const uint32_t p1 = (uint32_t)__popc(v[0]), p2 = p1 + (uint32_t)__popc(v[1]), p3 = p2 + (uint32_t)__popc(v[2]),
                         T = p3 + (uint32_t)__popc(v[3]);
          // valid points in the board's lanes up to this one (S) and from this one on (Q): four one-lane wave shifts each
          // way, cut at the board's first / last lane
          const uint32_t mlo = t5 >= 1 ? ~0u : 0u, mhi = t5 <= 3 ? ~0u : 0u;
          uint32_t S = T, Q = T;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            S = T + (dpp0<0x138>(S) & mlo);
            Q = T + (dpp0<0x130>(Q) & mhi);
          }
          const uint32_t P = S - T, n = S + Q - T;   // before this lane / on the whole board
          x = ((uint64_t)rngv[2 * sb + 1] << 32) | rngv[2 * sb];
          const uint64_t u = splitmix_next(x);
          const uint32_t k = (uint32_t)(((u >> 32) * (uint64_t)(n + 1)) >> 32);   // k == n: the pass
          const bool hit = k >= P && k < P + T;        // this lane holds the k-th valid point
          uint32_t tt = k - P, vr = v[0];
          int rr = 0;
          if (tt >= p1) { rr = 1; vr = v[1]; }
          if (tt >= p2) { rr = 2; vr = v[2]; }
          if (tt >= p3) { rr = 3; vr = v[3]; }
          tt -= rr == 0 ? 0u : (rr == 1 ? p1 : (rr == 2 ? p2 : p3));
#pragma unroll
          for (int sh = 16; sh >= 1; sh >>= 1) {   // the tt-th set bit of vr
            const uint32_t c = (uint32_t)__popc((vr >> pos) & ((1u << sh) - 1u));
            if (tt >= c) { tt -= c; pos += sh; }
          }
          rabs = r0 + rr;
          // -1: the board does not move this ply.  The lane with the point announces it; a pass / an idle board is
          // announced by the board's first lane
          a = !live ? -1 : (k < n ? rabs * N + (int)pos : hf.P);
          wr_act = bl && (live && k < n ? hit : t5 == 0);
          place = bl && live && hit;
        }
        if (wr_act) actv[sb] = a;
        if (!MOVES && bl && t5 == 0 && live) { rngv[2 * sb] = (uint32_t)x; rngv[2 * sb + 1] = (uint32_t)(x >> 32); }
        if (__ballot(live) == 0) break;
        resetm = __ballot(reset && t5 == 0);
        while (resetm) {   // rare
          const int s = ((__ffsll((unsigned long long)resetm) - 1) * 13) >> 6;   // lane 5 s -> board s
          resetm &= resetm - 1;
          for (int i = hf.lane; i < 5 * RS; i += kWave) st[(i / RS) * PL + s * RS + (i % RS)] = 0;
          if (hf.lane == 0) flagsv[s] = 8u;
        }
        WAVE_SYNC();
        // the new stone goes into the mover's plane right away: every later phase sees the position with it
        if (place) {
          const int turn = reset ? 0 : (int)(fl & 1u);
          st[turn * PL + sb * RS + rabs] |= 1u << pos;
        }
      }
      WAVE_SYNC();

      // phase 2 - one lane per (board, role): role 0 floods the mover's stones from the new stone q, roles 1-4 the
      // opponent's from the four neighbours of q; then every lane counts the liberties of its own group
      {
        const int s = (ln * 13) >> 6, j = ln - 5 * s;
        const bool used = ln < 5 * nb;
        const int ss = used ? s : 0;
        const int a = used ? actv[ss] : -1;
        const int turn = flagsv[ss] & 1u;
        const bool moving = a >= 0 && a < hf.P;
        int ar = -9, ac = 0;
        if (moving) split_action(a, N, hf.inv, ar, ac);
        const uint32_t bit = moving ? (1u << ac) : 0u;
        const uint32_t *own = st + ((j == 0) ? turn : 1 - turn) * PL + ss * RS;   // the colour this lane floods
        const uint32_t *oth = st + ((j == 0) ? 1 - turn : turn) * PL + ss * RS;
        const int sr = ar + (j == 1 ? -1 : (j == 2 ? 1 : 0));
        const uint32_t sbit = j == 3 ? (bit >> 1) : (j == 4 ? (bit << 1) : bit);
        uint32_t cnt = 0, sz = 0;
        {
          uint32_t m[R];
          {
            uint32_t mrev[R], f[R];
            uint32_t mt[RV * 4];
            const uint4 *pm = reinterpret_cast<const uint4 *>(own);
#pragma unroll
            for (int i = 0; i < RV; ++i) {
              const uint4 x = pm[i];
              mt[4 * i] = x.x; mt[4 * i + 1] = x.y; mt[4 * i + 2] = x.z; mt[4 * i + 3] = x.w;
            }
            // the seed is one bit of row sr: a one-hot row selector turns "r == sr" into a sign-extending bit extract
            // (the flood keeps its odd rows bit-reversed: their seeds are cut out of mrev with the reversed seed bit)
            const uint32_t onehot = (sr >= 0 && sr < R) ? (1u << sr) : 0u;
            const uint32_t sbit_rev = __brev(sbit);
#pragma unroll
            for (int r = 0; r < R; ++r) {
              m[r] = mt[r];
              mrev[r] = __brev(m[r]);
              const uint32_t sel = (uint32_t)__builtin_amdgcn_sbfe((int)onehot, r, 1);   // 0 or ~0
              f[r] = (r & 1) ? B3(mrev[r], sbit_rev, sel, TA & TB & TC) : B3(m[r], sbit, sel, TA & TB & TC);
            }
            flood2_serial<R, true>(m, mrev, f, sc + (used ? ln : 5 * kNB3) * RS);
          }
          // liberties of this lane's group on the position with the new stone (captures not yet removed); m[] still
          // holds the flooded colour's rows
          uint32_t gt[RV * 4], ot[RV * 4];
          const uint4 *pg = reinterpret_cast<const uint4 *>(sc + (used ? ln : 5 * kNB3) * RS);
          const uint4 *po = reinterpret_cast<const uint4 *>(oth);
#pragma unroll
          for (int i = 0; i < RV; ++i) {
            const uint4 x = pg[i], y = po[i];
            gt[4 * i] = x.x; gt[4 * i + 1] = x.y; gt[4 * i + 2] = x.z; gt[4 * i + 3] = x.w;
            ot[4 * i] = y.x; ot[4 * i + 1] = y.y; ot[4 * i + 2] = y.z; ot[4 * i + 3] = y.w;
          }
          const uint32_t fullrow = (1u << N) - 1u;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const uint32_t e = (FULLN || r < N) ? B3(ot[r], m[r], fullrow, ~(TA | TB) & TC & 0xFF) : 0u;   // empty points
            const uint32_t up = r > 0 ? gt[r - 1] : 0u, dn = r + 1 < R ? gt[r + 1] : 0u;   // rows >= R are not written
            const uint32_t d = B3(shl1(gt[r]), gt[r] >> 1, up, T_OR3);
            const uint32_t l = B3(d, dn, e, (TA | TB) & TC);
            cnt += (uint32_t)__popc(l);   // only min(cnt, 2) is used: one accumulating v_bcnt per row
            sz += gt[r];   // sum of the row words: equals the seed bit iff the group is the seed stone alone
          }
        }
        // roles 1-4: is this neighbour of q off the board or an opponent stone?  (all four: the new stone is boxed in)
        const bool off = j == 1 ? ar == 0 : (j == 2 ? ar == N - 1 : (j == 3 ? ac == 0 : ac == N - 1));
        const bool okbox = off || sz != 0u;   // the flooded group is non-empty iff its seed was a stone
        clsv[ln] = (cnt < 2u ? cnt : 2u) | ((sz != 0u && sz == sbit) ? 4u : 0u) | (sz != 0u ? 8u : 0u) | (okbox ? 16u : 0u);
        // an opponent group that keeps >= 2 liberties keeps its class: phase 3 must not see it
        if (j != 0 && cnt >= 2u) {
          uint4 *pz = reinterpret_cast<uint4 *>(sc + (used ? ln : 5 * kNB3) * RS);
#pragma unroll
          for (int i = 0; i < RV; ++i) pz[i] = make_uint4(0u, 0u, 0u, 0u);
        }
      }
      WAVE_SYNC();

      // phase 3 - all twelve boards in ONE pass, four adjacent rows per lane (lane -> board s5 = lane / 5, rows 4t .. 4t+3
      // with t = lane % 5): patch the classes, resolve captures and ko, the next mover's mask.  What does not depend on
      // the row (addresses, class decode, ballots, flags) is paid once per ply; with one row per lane and three boards per
      // pass it was paid four times (4.7e9 steps/s), with two rows and six boards twice (5.1e9).
      {
        const int sa = s5c;          // lanes 60-63 shadow board 0: always a real slot (< kNB3)
        const bool act = s5 < nb;    // (nb <= kNB3 = 12: lanes 60-63 have s5 = 12)
        const int av = actv[sa];
        const int a = act ? av : -1;
        const bool moves = a >= 0;
        const uint32_t fl = flagsv[sa];
        int turn = fl & 1u, passed = (fl >> 1) & 1u, done = (fl >> 2) & 1u;
        const uint32_t c0 = clsv[5 * sa], c1 = clsv[5 * sa + 1], c2 = clsv[5 * sa + 2], c3 = clsv[5 * sa + 3],
                       c4 = clsv[5 * sa + 4];
        // planes by role, not by colour: the mover's stones / classes are plane `turn` / 3 + `turn`
        uint4 *pmine = reinterpret_cast<uint4 *>(st + turn * PL + sa * RS + r0);
        uint4 *popp = reinterpret_cast<uint4 *>(st + (1 - turn) * PL + sa * RS + r0);
        uint4 *pMm = reinterpret_cast<uint4 *>(st + (3 + turn) * PL + sa * RS + r0);
        uint4 *pMo = reinterpret_cast<uint4 *>(st + (4 - turn) * PL + sa * RS + r0);
        const uint4 *gr = reinterpret_cast<const uint4 *>(sc + (5 * sa) * RS + r0);   // block j: gr[j * RS / 4]
        const bool rowt = act && r0 < R;     // the floods write rows 0 .. R-1 of their blocks only (rows >= R: full4 == 0)
        uint32_t mine1[4] = {0u, 0u, 0u, 0u}, opp0[4] = {0u, 0u, 0u, 0u}, Mm[4] = {0u, 0u, 0u, 0u}, Mo[4] = {0u, 0u, 0u, 0u},
                 g0[4] = {0u, 0u, 0u, 0u}, gch[4] = {0u, 0u, 0u, 0u};
        if (rowt) {
          unpack4(*pmine, mine1); unpack4(*popp, opp0); unpack4(*pMm, Mm); unpack4(*pMo, Mo);   // (rows >= N are zero)
          uint32_t g1[4], g2[4], g3[4], g4[4];
          unpack4(gr[0], g0); unpack4(gr[RS / 4], g1); unpack4(gr[2 * RS / 4], g2); unpack4(gr[3 * RS / 4], g3);
          unpack4(gr[RS], g4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            g0[r] &= full4[r];
            gch[r] = (g1[r] | g2[r] | g3[r] | g4[r]) & full4[r];   // the opponent groups whose class changes
          }
        }
        const bool is_pass = a == hf.P;
        const bool k1 = (c1 & 11u) == 8u, k2 = (c2 & 11u) == 8u, k3 = (c3 & 11u) == 8u, k4 = (c4 & 11u) == 8u;
        uint32_t cap[4] = {0u, 0u, 0u, 0u}, Mm_fix[4] = {0u, 0u, 0u, 0u};
        uint32_t libsG = c0 & 3u;   // libsG: liberties of G among the empty points (saturated at 2)
        int ko_r = -1, ko_c = 0;
        if (__ballot(moves && (k1 || k2 || k3 || k4))) {   // a capture on some board
          if (rowt) {
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            uint32_t g1[4], g2[4], g3[4], g4[4];
            unpack4(k1 ? gr[RS / 4] : z, g1); unpack4(k2 ? gr[2 * RS / 4] : z, g2); unpack4(k3 ? gr[3 * RS / 4] : z, g3);
            unpack4(k4 ? gr[RS] : z, g4);
#pragma unroll
            for (int r = 0; r < 4; ++r) cap[r] = (g1[r] | g2[r] | g3[r] | g4[r]) & full4[r];
          }
          // captured stones next to G are liberties of G too
          {
            uint32_t d[4];
            dilate_quad(g0, d);
            uint32_t any = 0, cnt = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const uint32_t x = d[r] & cap[r]; any |= x; cnt += (uint32_t)__popc(x); }
            const uint32_t nz = fifth_of(__ballot(any != 0), s5), many = fifth_of(__ballot(cnt > 1u), s5);
            libsG += ((nz & (nz - 1u)) | many) ? 2u : (nz ? 1u : 0u);
          }
          // gogame.py:72-75: ko iff exactly one stone died and the new stone is boxed in
          const uint32_t ncap1 = (k1 && (c1 & 4u) ? 1u : 0u) + (k2 && (c2 & 4u) ? 1u : 0u) + (k3 && (c3 & 4u) ? 1u : 0u) +
                                 (k4 && (c4 & 4u) ? 1u : 0u);
          const uint32_t ncapn = (k1 ? 1u : 0u) + (k2 ? 1u : 0u) + (k3 ? 1u : 0u) + (k4 ? 1u : 0u);
          const bool boxed = (c1 & c2 & c3 & c4 & 16u) != 0;
          if (moves && boxed && ncapn == 1u && ncap1 == 1u) {
            int ar, ac;
            split_action(a, N, hf.inv, ar, ac);
            ko_r = ar + (k1 ? -1 : (k2 ? 1 : 0));
            ko_c = ac + (k3 ? -1 : (k4 ? 1 : 0));
          }
          // the mover's groups in atari next to a captured stone (and not merged into G) now have >= 2 liberties
          uint32_t atari[4], f[4];
          dilate_quad(cap, f);
          uint32_t anyf = 0;
#pragma unroll
          for (int r = 0; r < 4; ++r) { atari[r] = mine1[r] & ~Mm[r] & ~g0[r]; f[r] &= atari[r]; anyf |= f[r]; }
          if (__ballot(anyf != 0)) {
#pragma unroll 1
            for (int it = 0; it < R * R; ++it) {
              uint32_t d[4], chg = 0;
              dilate_quad(f, d);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const uint32_t nw = B3(d[r], atari[r], f[r], T_ANDOR);
                chg |= nw ^ f[r];
                f[r] = nw;
              }
              if (__ballot(chg != 0) == 0) break;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Mm_fix[r] = f[r];
          }
        }
        const uint32_t gsel = libsG >= 2u ? ~0u : 0u;
        uint32_t Mo2[4], opp1[4], Mm2[4], e[4], x[4], nbr[4], invalid[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Mo2[r] = Mo[r] & ~gch[r];
          opp1[r] = opp0[r] & ~cap[r];
          Mm2[r] = B3(Mm[r], gsel, g0[r], (TA & ~TC & 0xFF) | (TB & TC)) | Mm_fix[r];   // (Mm & ~g0) | (gsel & g0)
          // state_utils.compute_invalid_moves on the lane's rows (invalid_from2, four rows per lane)
          e[r] = full4[r] & ~(opp1[r] | mine1[r]);
          x[r] = B3(e[r], opp1[r] & Mo2[r], mine1[r] & ~Mm2[r], T_OR3);
        }
        dilate_quad(x, nbr);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          invalid[r] = full4[r] & ~(e[r] & nbr[r]);
          if (r0 + r == ko_r) invalid[r] |= 1u << ko_c;
        }
        if (moves) {
          if (is_pass) { if (passed) done = 1; passed = 1; } else passed = 0;
          turn ^= 1;
          if (rowt) {
            *popp = make_uint4(opp1[0], opp1[1], opp1[2], opp1[3]);
            *reinterpret_cast<uint4 *>(st + 2 * PL + sa * RS + r0) = make_uint4(invalid[0], invalid[1], invalid[2], invalid[3]);
            *pMm = make_uint4(Mm2[0], Mm2[1], Mm2[2], Mm2[3]);
            *pMo = make_uint4(Mo2[0], Mo2[1], Mo2[2], Mo2[3]);
          }
          if (t5 == 0) {
            flagsv[sa] = (uint32_t)turn | ((uint32_t)passed << 1) | ((uint32_t)done << 2) | 8u;
            lastv[sa] = a;
            playedv[sa] += 1;
          }
        }
      }
      WAVE_SYNC();
    }

    // ---------------------------------------------------------------- store
    WAVE_SYNC();
    if (TRACKED) {
      const int64_t nbrd = (B - b_first) < nb ? (B - b_first) : nb;
      const int nw = (int)nbrd * W;
      uint32_t *gp = reinterpret_cast<uint32_t *>(states) + b_first * (int64_t)W;
      const uint32_t invW = ((1u << 20) + (uint32_t)W - 1u) / (uint32_t)W;   // i / W exactly for i < 12 * 96
      for (int i = hf.lane; i < nw; i += kWave) {
        const int sb = (int)(((uint32_t)i * invW) >> 20), w = i - sb * W;
        if (playedv[sb] == 0) continue;                 // untouched boards are not rewritten
        uint32_t v;
        if (w == 5 * N) {
          v = flagsv[sb] & 7u;
        } else {
          const int pl = (int)(((uint32_t)w * hf.inv) >> 16), rw = w - pl * N;
          v = st[pl * PL + sb * RS + rw];
        }
        gp[i] = v;
      }
      if (hf.lane < nb && b_first + hf.lane < B) {
        const int sb = hf.lane;
        const int64_t b = b_first + sb;
        const int played = playedv[sb];
        if (!MOVES) rng[b] = ((uint64_t)rngv[2 * sb + 1] << 32) | rngv[2 * sb];
        if (last_actions) last_actions[b] = lastv[sb];
        if (steps_done) steps_done[b] += played;
        if (MOVES && played_out) played_out[b] = played;
      }
      WAVE_SYNC();
    }
    if (IO == 0) load_spread_lut(lut, hf.lane);
#pragma unroll 1
    for (int i = 0; i < (TRACKED ? 0 : nb / 2); ++i) {
      const int s = 2 * i + hf.h;
      const uint32_t fl = flagsv[s];
      const bool on = (fl >> 3) & 1u;
      const int64_t b = on ? b_first + s : B - 1;
      const int played = playedv[s];
      uint32_t black = 0, white = 0, invalid = 0;
      if (row) {
        black = st[0 * PL + s * RS + hf.hl];
        white = st[1 * PL + s * RS + hf.hl];
        invalid = st[2 * PL + s * RS + hf.hl];
      }
      const bool wr = on && played != 0;
      if (TRACKED) {
        uint32_t *gp = reinterpret_cast<uint32_t *>(states) + b * (int64_t)W;
        if (wr && hf.hl < N) {
          gp[hf.hl] = black; gp[N + hf.hl] = white; gp[2 * N + hf.hl] = invalid;
          gp[3 * N + hf.hl] = st[3 * PL + s * RS + hf.hl];
          gp[4 * N + hf.hl] = st[4 * PL + s * RS + hf.hl];
        }
        if (wr && hf.hl == 31) gp[5 * N] = fl & 7u;
      } else if (PACKED) {
        store_packed_h(reinterpret_cast<uint32_t *>(states) + b * (int64_t)W, N, hf, black, white, invalid, fl & 1u,
                       (fl >> 1) & 1u, (fl >> 2) & 1u, wr);
      } else if (__ballot(wr)) {
        emit_store_h<R>(states + b * (int64_t)S, black, white, invalid, fl & 1u, (fl >> 1) & 1u, (fl >> 2) & 1u, hf,
                        v2 + hf.h * 128, lut, wr);
      }
      if (on && hf.hl == 0) {
        if (!MOVES) rng[b] = ((uint64_t)rngv[2 * s + 1] << 32) | rngv[2 * s];
        if (last_actions) last_actions[b] = lastv[s];
        if (steps_done) steps_done[b] += played;
        if (MOVES && played_out) played_out[b] = played;
      }
      WAVE_SYNC();
    }
  }
}

// byte planes -> tracked boards: the rows of planes 0 / 1 / 3 and the liberty classes of one v2 analysis
template <int R>
__global__ __launch_bounds__(kWave, 4) void k_track(const uint8_t *__restrict__ states, uint32_t *__restrict__ tracked,
                                                     int64_t B, int N, uint32_t inv) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[Lds2<R>::kTotal];
  const Half hf = make_half(threadIdx.x, N, inv);
  load_cw_table<R>(lds, hf.lane);
  const int S = 6 * hf.P, W = 5 * N + 1;
  uint8_t *io = reinterpret_cast<uint8_t *>(lds) + hf.h * Cfg<R>::kIoBytes;
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
