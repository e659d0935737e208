// gg_kernels.hip - C-ABI (include/gymgo_amd.h) of the MI355X batched Go step path: argument checks, device selection,
// grid sizing and dispatch on the board-size template.  The kernels live in gg_common.h (shared building blocks),
// gg_v2.h (per-ply kernels: two boards per wavefront, every liberty class from scratch), gg_v4.h (multi-ply kernels:
// sixteen boards per wavefront, liberty classes carried from ply to ply), gg_aux.h (stand-alone sampler and capture
// resolution), gg_ws.h (policy-weighted sampling), gg_sym.h (batched symmetries) and gg_ns16.h (the per-ply kernels for
// big batches).  Which kernel serves an entry point depends on the arguments only (board size, batch size, plies per
// launch) and on the CU count the grids are sized for - the device's own, or GYMGO_AMD_CUS (forced_cus below), the one
// environment variable the shipped build reads; results depend on neither.  Mutable global state, all of it performance-only
// (no result depends on any of it): g_cus (CU count per device, relaxed atomics: racing first callers store the same value), the
// occupancy cache of waves_per_simd_of (per device and kernel, behind a mutex) and the FairShare progress board in device
// memory (gg_common.h: one word per hardware wave slot, written by every fused launch; foreign or stale entries only shift
// issue priorities).  Every entry point is re-entrant and may be called from several threads on several streams.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <stdlib.h>   // getenv: GYMGO_AMD_CUS (below); the GG_AB_* tuning overrides exist in A/B builds (make ab) only
#ifdef GG_AB
#include <stdio.h>
#endif

#include "gg_common.h"
#include "gg_v2.h"
#include "gg_v4.h"
#include "gg_v5.h"
#include "gg_aux.h"
#include "gg_ws.h"
#include "gg_sym.h"
#include "gg_ns16.h"
#include "gymgo_amd.h"

namespace gg {
// gg_rollout.hip: the fused multi-ply launches with drawn moves, a translation unit of their own (one code-generation switch differs)
void launch_rollout4(int io, uint8_t *st, uint64_t *rng, int32_t *last_actions, int64_t *steps_done, int64_t B, int32_t N,
                     uint32_t inv, int plies, int auto_reset, int nb, int grid, hipStream_t s);
void launch_rollout5(int io, int N, uint8_t *st, uint64_t *rng, int32_t *last_actions, int64_t *steps_done, int64_t B, uint32_t inv,
                     int plies, int auto_reset, int nb, int grid, hipStream_t s);
void launch_rollout_lat(int io, uint8_t *st, uint64_t *rng, int32_t *last_actions, int64_t *steps_done, int64_t B, int32_t N, int plies,
                        int auto_reset, bool w4, hipStream_t s);
void launch_env_step_lat(uint32_t *tracked, uint64_t *rng, int64_t *steps_done, int64_t B, int32_t N, int auto_reset,
                         const EnvArgs &env, bool w4, hipStream_t s);
}

namespace {

using namespace gg;

constexpr int kMaxDevices = 64;
std::atomic<int> g_cus[kMaxDevices];   // 0 = not queried yet; racing first callers all store the same value (relaxed atomics:
                                       // round 4's ThreadSanitizer pass flagged the plain ints this used to be)

// GYMGO_AMD_CUS=<n> (read once per process): size every grid, batch split and kernel choice for n compute units instead of
// the device's own count - a partition of the GPU (CPX / TPX), a GPU shared with other work, and tests/test_gpu_cus.py, which
// runs the entry points for 8 ... 1 024 units and expects bit-identical results (no kernel waits for a co-resident workgroup,
// so a grid larger than the device is merely queued).
int forced_cus() {
  static const int forced = [] {
    const char *e = getenv("GYMGO_AMD_CUS");
    if (!e || !*e) return 0;
    char *end = nullptr;
    const long v = strtol(e, &end, 10);   // the WHOLE string must be a number: "12abc" or "8 " is a typo, not 12 or 8 - ignored
    return (end && *end == '\0' && v > 0 && v <= 4096) ? (int)v : 0;
  }();
  return forced;
}

int cus_of(int dev) {
  if (const int f = forced_cus()) return f;
  if (dev < 0 || dev >= kMaxDevices) return 256;
  int c = g_cus[dev].load(std::memory_order_relaxed);
  if (c <= 0) {
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) return 256;
    g_cus[dev].store(c, std::memory_order_relaxed);
  }
  return c;
}

// Kernels are launched on the device that OWNS the buffers (the stream handed over belongs to it too), whatever the
// calling thread's current device is; the current device is restored on return.  One hipPointerGetAttributes per call.
struct OnDeviceOf {
  int prev = -1, dev = 0;
  bool switched = false;
  explicit OnDeviceOf(const void *p) {
    (void)hipGetDevice(&prev);
    dev = prev < 0 ? 0 : prev;
    hipPointerAttribute_t at;
    if (p && hipPointerGetAttributes(&at, p) == hipSuccess) dev = at.device;
    else (void)hipGetLastError();   // a pointer HIP does not know: launch on the current device (and fail there)
    if (dev != prev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~OnDeviceOf() {
    if (switched) (void)hipSetDevice(prev);
  }
  int cus() const { return cus_of(dev); }
};

// persistent grid: enough single-wave workgroups to fill every SIMD several times over
int grid_for(int cus, int64_t work, int per_cu = 32) {
  int64_t cap = (int64_t)cus * per_cu;
#ifdef GG_AB
  if (const char *e = getenv("GG_AB_GRID_CAP")) cap = atoll(e);
#endif
  return (int)(work < cap ? (work > 0 ? work : 1) : cap);
}

// pipelined kernels: exactly the resident set (waves per SIMD x 4 SIMDs x CUs), so that each workgroup runs many
// iterations and pays the un-overlapped first load / last store once
int grid_resident(int cus, int64_t work, int waves_per_simd) {
  int64_t cap = (int64_t)cus * 4 * waves_per_simd;
  return (int)(work < cap ? (work > 0 ? work : 1) : cap);
}

// Persistent per-pair kernels (pair_span, gg_common.h): from two pairs per resident wave on, the grid is exactly the
// resident set of THIS kernel (its occupancy, asked of the runtime once per kernel) and the pairs of a SIMD are split by
// wave age.  Cumulative shares from sweeps on 65 536 boards (tools/exp/age_split.py (round 3, in git history); 32 pairs per SIMD): three waves
// 0.40 / 0.74 (12 / 11 / 9 pairs: gg_batch_env_step 81.0 -> 73 us, the one-ply gg_batch_rollout 77.4 -> 69.5 us), four
// waves 0.39 / 0.665 / 0.86 (12 / 9 / 6 / 5: invalid mask 50.9 -> 47, track 50.4 -> 45, packed next states 42.8 -> 41,
// packed env step 52.3 -> 47.5 us).  Equal shares on the same resident grid gain nothing or lose (tools/exp/grid_cap.py (round 3, in git history)).
// Anything smaller, or an occupancy the split has no shares for: one wave per pair, at most 32 per CU, as before.
// (the occupancy is a fact of the CURRENT device - OnDeviceOf has made the buffers' device current by now - so the cache
// is keyed on (device, kernel): a process that drives devices of different SKUs or partition modes gets each one's own)
int waves_per_simd_of(const void *kern) {
  static std::mutex mu;
  static std::unordered_map<uint64_t, int> known;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  const uint64_t key = (uint64_t)(uintptr_t)kern * 64u + (uint64_t)(dev & 63);   // (kernel stubs are >= 16 bytes apart)
  std::lock_guard<std::mutex> lock(mu);
  auto it = known.find(key);
  if (it != known.end()) return it->second;
  int blocks = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kern, kWave, 0) != hipSuccess) { (void)hipGetLastError(); blocks = 0; }
  const int w = blocks / 4;
  known.emplace(key, w);
  return w;
}

AgeSplit age_split(const void *kern, int cus, int64_t npairs, bool split, int &grid) {
  AgeSplit as = {0, {0, 0, 0}};
  grid = grid_for(cus, npairs);
  int w = split ? waves_per_simd_of(kern) : 0;
  if (w < 3 || w > kAgeRanks) w = 0;   // measured: three and four waves per SIMD
  // the share of the wave of age rank r (oldest first).  More than four waves per SIMD (9x9 boards: 5-7) keep equal
  // shares: a geometric fall-off extrapolated from these was measured slower than even (9x9 x 65 536: 51.3 vs 49.6 us)
  static const double kShare[kAgeRanks + 1][kAgeRanks] = {
      {0}, {0}, {0}, {0.40, 0.34, 0.26}, {0.39, 0.275, 0.195, 0.14}};
  double c[kAgeRanks - 1];
  double acc = 0;
  for (int i = 0; i < kAgeRanks - 1; ++i) {
    acc += i < w ? kShare[w][i] : 0.0;
    c[i] = i < w - 1 ? acc : 1.0;
  }
#ifdef GG_AB
  if (getenv("GG_AB_EVEN")) w = 0;
  char name[16];
  snprintf(name, sizeof name, "GG_AB_CUT%d", w);
  if (const char *e = getenv(name)) sscanf(e, "%lf,%lf,%lf", &c[0], &c[1], &c[2]);
#endif
  if (w && npairs >= (int64_t)cus * 4 * w * 2) {
    as.cols = cus * 4;
    for (int i = 0; i < kAgeRanks - 1; ++i) as.cut[i] = (uint32_t)(c[i] * 65536.0);
    grid = as.cols * w;
  }
  return as;
}

// Grid of a sixteen-board kernel (gg_ns16.h).  19x19: the resident set of three waves per SIMD with a SIMD's groups split
// by wave age (cumulative shares c1, c2) - but only when this kernel really has three resident waves per SIMD on this
// device; any other occupancy (a compiler that needs more registers, a partitioned device) takes one workgroup per group
// like the smaller boards, whose results are the same and whose performance degrades gently.
template <typename... KArgs>
AgeSplit ns16_grid(void (*kern19)(KArgs...), int cus, int64_t ngroups, int32_t N, uint32_t c1, uint32_t c2, int &grid16) {
  AgeSplit as = {0, {c1, c2, 65536u}};
  grid16 = (int)ngroups;
  if (N == 19 && waves_per_simd_of(reinterpret_cast<const void *>(kern19)) == 3) {
    as.cols = cus * 4;
    grid16 = as.cols * 3;
  }
  return as;
}

template <typename... KArgs, typename... Args>
void launch_pairs(void (*kern)(KArgs...), int cus, int64_t npairs, bool split, hipStream_t s, Args... args) {
  int grid;
  const AgeSplit as = age_split(reinterpret_cast<const void *>(kern), cus, npairs, split, grid);
  // (the env-step kernels take two more trailing arguments - the areas output and the reward formula of
  // gg_batch_env_step_scored - which every other caller leaves at "none"; function pointers carry no default arguments)
  if constexpr (sizeof...(KArgs) == sizeof...(Args) + 3) kern<<<grid, kWave, 0, s>>>(args..., as, nullptr, 0);
  else kern<<<grid, kWave, 0, s>>>(args..., as);
}

// The sixteen-boards-per-wave kernels of gg_ns16.h serve batches of exactly 9x9 / 13x13 / 19x19 boards from a number of
// groups per SIMD on that depends on the entry point and the board size (tools/exp/ns16_min.py (round 3, in git history), us per call, two-board
// kernel -> sixteen-board kernel):
//   gg_batch_next_states    19x19 from 4 (49 152 boards: 48.6 -> 47.3, 32 768: 35.7 -> 38.3), 13x13 and 9x9 from 2
//                           (32 768: 28.7 -> 27.6 / 23.9 -> 18.7; 16 384: 18.5 -> 19.2 / 14.2 -> 15.1)
//   gg_batch_env_step       19x19 from 3 (49 152: 57.7 -> 52.3, 32 768: 42.6 -> 41.0, 24 576: 35.3 -> 39.1), 13x13 from 2
//   (+ one-ply rollout)     (32 768: 32.5 -> 26.7, 24 576: 24.1 -> 25.5), 9x9 from 1 (16 384: 17.1 -> 12.8)
//   gg_batch_invalid_mask   19x19 from 4 (65 536: 49.0 -> 47.0, 49 152: 37.0 -> 38.5), 13x13 from 2 (32 768: 20.8 -> 17.5),
//                           9x9 from 1 (16 384: 10.0 -> 9.0)
//   gg_batch_track_states   like the mask (round 4: k_track16)
bool use_ns16(int cus, int64_t B, int32_t N, int per19, int per13, int per9) {
  const int64_t ngroups = (B + 15) / 16;
  int per_simd = N == 19 ? per19 : N == 13 ? per13 : per9;
  bool big = (N == 9 || N == 13 || N == 19) && per_simd > 0 && ngroups <= 0x7FFFFFFF;
#ifdef GG_AB
  if (const char *e = getenv("GG_AB_NS16")) big = big && atoi(e) != 0;
  if (const char *e = getenv("GG_AB_NS16_MIN")) per_simd = atoi(e);
#endif
  return big && ngroups >= (int64_t)cus * 4 * per_simd;
}

// multi-ply kernel: boards per wave (even, <= kNB4 = 16).  The flood batch of a ply costs the same for 2 or 16 boards, so
// the more the better; small batches take fewer per wave so that every SIMD still gets a wave.  65 536 games on 256 CUs:
// 16 boards x 4 096 waves = exactly the resident set (4 waves per SIMD).
int boards_per_wave(int cus, int64_t B, int &grid) {
  int64_t nb = B / ((int64_t)cus * 4);
#ifdef GG_AB
  if (const char *e = getenv("GG_AB_NB")) nb = atoi(e);
#endif
  if (nb > kNB4) nb = kNB4;
  nb &= ~(int64_t)1;
  if (nb < 2) nb = 2;
  grid = grid_for(cus, (B + nb - 1) / nb);
  return (int)nb;
}

// Byte-plane / packed boards enter a multi-ply launch through one full analysis per board; the multi-ply kernel pays
// off once every SIMD can get a wave of >= 8 boards and from two plies per launch on (round-1 measurements: 9x9 x 4 096
// games 1.4e9 vs 2.0e9 steps/s on the per-ply kernel, 16 384 games on par, 262 144 games 9.7e9 vs 2.9e9; 65 536 games at
// 2 / 3 / 5 plies per launch 1.31 / 1.78 / 2.47e9 against 1.14 / 1.32 / 1.51e9).  Tracked boards carry their classes
// and always run it.
bool use_multi_ply(int cus, int64_t B, int plies) {
  int64_t min_games = (int64_t)cus * 32;
#ifdef GG_AB
  if (const char *e = getenv("GG_AB_MULTI_MIN")) min_games = atoll(e);
#endif
  return plies >= 2 && B >= min_games;
}

// The thirty-two-board multi-ply kernel (gg_v5.h: a pair of lanes per board, the floods of a ply as a compacted job list) serves
// the fused launches of full-size 19x19 batches from the point on where k_rollout4 would need a THIRD wave per SIMD (more than 128
// games per CU; round 6, last session: also 9x9 and 13x13 batches, from 160 games per CU); 19.5 KB of LDS per wave = eight waves per CU = two per SIMD.  Boards per wave: as many as keep the rounds of
// resident waves full (65 536 games on 256 CUs: 32 boards x 2 048 waves; 34 816: 18 boards).  Measured new / k_rollout4, ms per
// launch of 256 plies (profiles/r06f_r5_edges.txt, r06f_r5_time.txt): 32 768 games 1.195 / 1.190, 34 816 1.208 / 1.455, 49 152
// 1.247 / 1.490, 65 536 1.304 / 1.835, 98 304 2.47 / 3.55, 131 072 2.57 / 3.62; and per launch length at 65 536 games: 2 plies
// 0.082 / 0.072, 4 0.093 / 0.088, 6 0.105 / 0.105, 8 0.114 / 0.122, 32 0.239 / 0.296 -> from 8 plies per launch on.
// (A/B builds: GG_AB_R5 = 0 / 1 forces the choice, GG_AB_R5_MIN = games per CU, GG_AB_R5_PLIES, GG_AB_NB5 = boards per wave.)
bool use_rollout5(int cus, int64_t B, int32_t N, int plies) {
  // 9x9 / 13x13 (tools/exp/r5_small.py, ms per launch of 256 plies, new / k_rollout4 or k_rollout_lat): 9x9 32 768 games 0.735 / 0.726,
  // 40 960 0.826 / 0.854, 49 152 0.845 / 0.869, 65 536 0.874 / 1.032, 131 072 1.98 / 2.36; 13x13 32 768 0.966 / 0.894, 40 960 0.987 /
  // 1.084, 49 152 1.009 / 1.104, 65 536 1.051 / 1.337, 131 072 2.32 / 2.64 -> from 160 games per CU on (13x13 with a row stride of
  // 20 words: with the 16 of the other kernels the blocks of every fourth board share their LDS banks, 1.42 ms at 65 536 games)
  int64_t per_cu = N == 19 ? 4 * kNB5 : 5 * kNB5 - 1;
  int min_plies = 8;
  bool ok = N == 19 || N == 13 || N == 9;
#ifdef GG_AB
  if (const char *e = getenv("GG_AB_R5_MIN")) per_cu = atoll(e);
  if (const char *e = getenv("GG_AB_R5_PLIES")) min_plies = atoi(e);
  if (const char *e = getenv("GG_AB_R5")) { if (atoi(e) == 0) ok = false; else { per_cu = -1; min_plies = 1; } }
#endif
  return ok && plies >= min_plies && B > (int64_t)cus * per_cu;
}
int boards_per_wave5(int cus, int64_t B, int &grid) {
  const int64_t resident = (int64_t)cus * 8;
  const int64_t waves = (B + kNB5 - 1) / kNB5, rounds = (waves + resident - 1) / resident;
  int64_t nb = (B + rounds * resident - 1) / (rounds * resident);
#ifdef GG_AB
  if (const char *e = getenv("GG_AB_NB5")) nb = atoi(e);
#endif
  nb = (nb + 1) & ~(int64_t)1;
  if (nb > kNB5) nb = kNB5;
  if (nb < 2) nb = 2;
  grid = grid_for(cus, (B + nb - 1) / nb);
  return (int)nb;
}

// The latency-shaped multi-ply kernel (gg_lat.h: one row per lane, four 9x9 / 13x13 boards or two 19x19 boards per wave, the
// ply in registers) serves the fused launches of batches that leave the SIMDs under-filled.  Take-over points measured per
// board size (profiles/r05a_lat_sweep.txt, A/B build, 256 plies per launch, new / old kernel):
//   9x9 (and smaller)  4 096 games x1.77, 8 192 x1.69, 16 384 x1.10, 32 768 x0.73     -> up to 64 games per CU
//   13x13 (10 .. 13)   4 096 x1.50, 8 192 x1.39, 16 384 x0.87                          -> up to 32 games per CU
//   19x19 (14 .. 19)   1 024 x1.07, 2 048 x1.05, 4 096 x0.89                           -> up to 8 games per CU (round 5; see below)
// and per launch length at 4 096 games: 9x9 1 / 2 / 4 / 16 plies x0.87 / 0.97 / 1.13 / 1.49, 13x13 x0.78 / 0.87 / 1.03 / 1.32,
// 19x19 x0.50 / 0.59 / 0.73 / 0.82 (the launch pays the first classes of every board: eleven lock-step floods), so from
// 3 / 3 - 4 / 8 plies per launch on.  (A/B builds: GG_AB_LAT_MAX = games per CU, GG_AB_LAT_PLIES = plies.)
// Tracked boards carry their classes - no first analysis on either kernel, a lane loads and stores its own five row words - so
// the kernel pays from ONE ply per launch on and up to larger batches (profiles/r05e_lat_tracked_sweep.txt, new / k_rollout4
// at 1 / 4 / 64 / 256 plies per launch): 9x9 4 096 games x1.06 / 2.06 / 2.33 / 2.54, 16 384 x1.61 / 1.41 / 1.16 / 1.14, 32 768
// x1.35 / 1.04 / 0.76 / 0.76; 13x13 8 192 x1.77 / 1.74 / 1.55 / 1.61, 16 384 x1.86 / 1.37 / 1.00 / 1.01; 19x19 4 096 x1.65 / 1.27 /
// 1.11 / 1.14, 8 192 x1.52 / 1.03 / 0.76 / 0.72 -> up to 64 / 64 / 16 games per CU.  (A/B builds: GG_AB_LATT_MAX, GG_AB_LATT_PLIES.)
bool use_lat(int cus, int64_t B, int32_t N, int plies, bool tracked = false) {
  // (round 6, three floods per ply instead of five - profiles/r06c_mid_batch.txt, 19x19 x 256 plies, new / best other family:
  // 2 048 games 0.414 / 0.637 ms, 4 096 0.557 / 0.736, 6 144 0.713 / 0.889, 8 192 0.874 / 0.888 (k_rollout4, eight boards per
  // wave), 12 288 1.31 / 0.95 -> up to 31 games per CU; the launch time is now monotone in the batch size)
  // (9x9 / 13x13 with the three-flood ply, profiles/r06c_mid_batch_9_13.txt, new / k_rollout4: 9x9 16 384 games 0.42 /
  // 0.64 ms, 24 576 0.58 / 0.73, 32 768 0.735 / 0.746, 49 152 1.04 / 0.89; 13x13 16 384 0.58 / 0.75, 24 576 0.90 / 0.90, 32 768
  // 1.11 / 0.92 -> up to 128 / 80 games per CU)
  int64_t per_cu = N <= 9 ? 128 : N <= 13 ? 80 : 31;
  // (second sweep, with every read of either kernel's prologue in flight together - profiles/r05n_lat_fsweep.txt, hipGraph nodes,
  // new / two-board: 9x9 4 096 games 1 / 2 / 3 / 4 / 16 plies x0.86 / 0.98 / 1.12 / 1.22 / 1.66; 13x13 4 096 x0.81 / 0.95 / 1.09 /
  // 1.19 / 1.57, 1 024 x0.71 / 0.82 / 0.93 / 1.01 / 1.33; 19x19 2 048 games 4 / 8 / 16 / 64 plies x0.92 / 1.04 / 1.12 / 1.19)
  int min_plies = N <= 9 ? 3 : N <= 13 ? (B >= (int64_t)cus * 16 ? 3 : 4) : 8;
  if (tracked) {
    per_cu = N <= 13 ? 64 : 16;
    min_plies = 1;
    // ONE ply per launch is a latency chain that many small waves overlap better than sixteen-board ones, at every batch size
    // measured (9x9 65 536 / 262 144 games x1.14 / 1.12, 13x13 32 768 / 131 072 x1.53 / 1.23, 19x19 16 384 / 32 768 / 65 536
    // x1.34 / 1.03 / 0.81; two plies per launch: x0.94 - 1.03 up to 13x13, 0.59 - 0.99 at 19x19)
    if (plies == 1) per_cu = N <= 13 ? (int64_t)1 << 40 : 128;
  }
#ifdef GG_AB
  if (const char *e = getenv(tracked ? "GG_AB_LATT_MAX" : "GG_AB_LAT_MAX")) per_cu = atoll(e);
  if (const char *e = getenv(tracked ? "GG_AB_LATT_PLIES" : "GG_AB_LAT_PLIES")) min_plies = atoi(e);
#endif
  return plies >= min_plies && B <= (int64_t)cus * per_cu;
}

// ... and the tracked env step (k_env_step_lat: one ply + GoEnv.step's outputs + the observation).  A one-ply launch is a
// latency chain, and more, smaller waves overlap their loads / plies / stores better than sixteen-board waves: measured new /
// k_rollout4 (profiles/r05k_lat_env_sweep.txt, with the observation): 9x9 4 096 games x1.18, 16 384 x1.43, 65 536 x1.06,
// 262 144 x1.03; 13x13 4 096 x1.40, 16 384 x1.69, 131 072 x1.22; 19x19 4 096 x1.62, 16 384 x1.38, 32 768 x1.11 (without the
// observation x0.96), 65 536 x0.98 (x0.78) -> every batch up to 13x13; 19x19 up to 128 games per CU with the observation, 64
// without.  (A/B builds: GG_AB_LATE_MAX = games per CU.)
bool use_lat_env(int cus, int64_t B, int32_t N, bool with_observation) {
  int64_t per_cu = N <= 13 ? (int64_t)1 << 40 : (with_observation ? 128 : 64);
#ifdef GG_AB
  if (const char *e = getenv("GG_AB_LATE_MAX")) per_cu = atoll(e);
#endif
  return B <= (int64_t)cus * per_cu;
}

// Short launches of the one-row-per-lane kernels on tracked boards go out as four-wave workgroups: single-wave workgroups
// enter the machine over ~0.26 ns each (tools/exp/oneply_ramp.py), a tenth of a one-ply launch of a thousand of them.
// (A/B builds: GG_AB_WPB = 1 / 4 forces the form.)
// hipGraph node, single-wave -> four-wave workgroups (profiles/r05p_wpb_tracked.txt): 9x9 x 4 096 games one-ply rollout 3.45 -> 3.10
// us, env step with / without the observation 4.70 -> 4.44 / 3.79 -> 3.52; 16 384 games 5.34 -> 4.80, 7.57 -> 7.17; 13x13 x 4 096 3.99
// -> 3.65; 5x5 / 7x7 / 11x11 alike; four plies per launch still -0.3 us; from 65 536 games of 9x9 on and at 19x19 nothing moves.
bool lat_w4(int cus, int64_t B, int32_t N, int plies) {
  const int64_t groups = (B + 3) / 4;
  bool w4 = N <= 13 && plies <= 4 && groups <= (int64_t)cus * 16;
#ifdef GG_AB
  if (const char *e = getenv("GG_AB_WPB")) w4 = atoi(e) == 4;
#endif
  return w4;
}

int32_t check(int64_t B, int32_t N) { return (N < 2 || N > GG_MAX_BOARD || B < 0) ? GG_E_BADSIZE : 0; }

// reciprocal for the in-kernel action -> (row, col) split; exactness is verified for every action
uint32_t recip16(int32_t N) {
  uint32_t inv = (65536u + (uint32_t)N - 1u) / (uint32_t)N;
  for (uint32_t a = 0; a <= (uint32_t)(N * N); ++a)
    if (((a * inv) >> 16) != a / (uint32_t)N) return 0;
  return inv;
}

// multi-ply kernel: the instantiation with compile-time N when the board fills its row capacity (9, 13, 19)
#define GG_DISPATCH4(N, IO, MOVES, GRID, ...)                                                          \
  do {                                                                                                  \
    if ((N) == 9) { k_rollout4<9, IO, MOVES, true><<<GRID, kWave, 0, s>>>(__VA_ARGS__); }        \
    else if ((N) < 9) { k_rollout4<9, IO, MOVES, false><<<GRID, kWave, 0, s>>>(__VA_ARGS__); }   \
    else if ((N) == 13) { k_rollout4<13, IO, MOVES, true><<<GRID, kWave, 0, s>>>(__VA_ARGS__); } \
    else if ((N) < 13) { k_rollout4<13, IO, MOVES, false><<<GRID, kWave, 0, s>>>(__VA_ARGS__); } \
    else if ((N) == 19) { k_rollout4<19, IO, MOVES, true><<<GRID, kWave, 0, s>>>(__VA_ARGS__); } \
    else { k_rollout4<19, IO, MOVES, false><<<GRID, kWave, 0, s>>>(__VA_ARGS__); }               \
  } while (0)

// the env-step instantiation (tracked boards, one ply, GoEnv.step outputs)
#define GG_DISPATCH4E(N, MOVES, GRID, ...)                                                                \
  do {                                                                                                     \
    if ((N) == 9) { k_rollout4<9, 2, MOVES, true, true><<<GRID, kWave, 0, s>>>(__VA_ARGS__); }        \
    else if ((N) < 9) { k_rollout4<9, 2, MOVES, false, true><<<GRID, kWave, 0, s>>>(__VA_ARGS__); }   \
    else if ((N) == 13) { k_rollout4<13, 2, MOVES, true, true><<<GRID, kWave, 0, s>>>(__VA_ARGS__); } \
    else if ((N) < 13) { k_rollout4<13, 2, MOVES, false, true><<<GRID, kWave, 0, s>>>(__VA_ARGS__); } \
    else if ((N) == 19) { k_rollout4<19, 2, MOVES, true, true><<<GRID, kWave, 0, s>>>(__VA_ARGS__); } \
    else { k_rollout4<19, 2, MOVES, false, true><<<GRID, kWave, 0, s>>>(__VA_ARGS__); }               \
  } while (0)

// ... with the moves drawn from policy weights by the kernel
#define GG_DISPATCH4W(N, GRID, ...)                                                                        \
  do {                                                                                                     \
    if ((N) == 9) { k_rollout4<9, 2, true, true, true, true><<<GRID, kWave, 0, s>>>(__VA_ARGS__); }        \
    else if ((N) < 9) { k_rollout4<9, 2, true, false, true, true><<<GRID, kWave, 0, s>>>(__VA_ARGS__); }   \
    else if ((N) == 13) { k_rollout4<13, 2, true, true, true, true><<<GRID, kWave, 0, s>>>(__VA_ARGS__); } \
    else if ((N) < 13) { k_rollout4<13, 2, true, false, true, true><<<GRID, kWave, 0, s>>>(__VA_ARGS__); } \
    else if ((N) == 19) { k_rollout4<19, 2, true, true, true, true><<<GRID, kWave, 0, s>>>(__VA_ARGS__); } \
    else { k_rollout4<19, 2, true, false, true, true><<<GRID, kWave, 0, s>>>(__VA_ARGS__); }               \
  } while (0)

#define GG_DISPATCH(N, CALL9, CALL13, CALL19) \
  do {                                        \
    if ((N) <= 9) { CALL9; }                  \
    else if ((N) <= 13) { CALL13; }           \
    else { CALL19; }                          \
  } while (0)

// GG_K(R, F) = the launch of the kernel for row capacity R; F: the board fills the capacity (N == R is a compile-time
// constant in that instantiation)
#define GG_DISPATCH_N(N)                     \
  do {                                       \
    if ((N) == 9) { GG_K(9, true); }         \
    else if ((N) < 9) { GG_K(9, false); }    \
    else if ((N) == 13) { GG_K(13, true); }  \
    else if ((N) < 13) { GG_K(13, false); }  \
    else if ((N) == 19) { GG_K(19, true); }  \
    else { GG_K(19, false); }                \
  } while (0)

}  // namespace

extern "C" {

#ifdef GG_AB_PROF
// A/B builds only: read and clear the phase clocks of THIS translation unit's launches (k_rollout2, env steps of k_rollout4;
// gg_prof has internal linkage; [8] / [9] = first entry / last exit on the 100 MHz wall clock)
GG_PROF_READ(gg_ab_prof_read_kernels)
GG_PROF_RAW(gg_ab_prof_raw_kernels)
#endif

#ifdef GG_AB_WHERE
int32_t gg_ab_where_read(unsigned int *out, int n) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(gg::gg_where), sizeof(unsigned int) * 3 * n) == hipSuccess ? 0 : 2;
}
#endif

int32_t gg_version(void) { return GG_ABI_VERSION; }

int32_t gg_device_cus(void) {
  if (const int f = forced_cus()) return f;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  int c = 0;
  if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return c;
}

// common prologue: size checks, empty batch, the (exact) reciprocal of N, the owning device, the stream
#define GG_ENTER(PTR)                                \
  if (int32_t e_ = check(B, N)) return e_;           \
  if (B == 0) return 0;                              \
  const uint32_t inv = recip16(N);                   \
  if (!inv) return GG_E_BADSIZE;                     \
  if (!(PTR)) return GG_E_NULLPTR;                   \
  OnDeviceOf on_dev(PTR);                            \
  const int cus = on_dev.cus();                      \
  hipStream_t s = (hipStream_t)hip_stream;           \
  (void)cus; (void)inv

int32_t gg_batch_next_states(const uint8_t *in, const int32_t *actions, uint8_t *out, int32_t *status, int64_t B,
                             int32_t N, int32_t canonical, void *hip_stream) {
  GG_ENTER(in);
  if (!actions || !out) return GG_E_NULLPTR;
  // Big batches of full-size boards: sixteen boards per wave, floods class-major (gg_ns16.h; use_ns16 above has the
  // batch sizes).  19x19 runs the resident set with a SIMD's groups split 2 : 1 : 1 by wave age (58.6 us per 65 536 boards
  // against 60.0 with one workgroup per group; 1 : 1 : 2 70.6), the smaller boards (four waves per SIMD) one workgroup
  // per group.
  {
    const int64_t ngroups = (B + kNB16 - 1) / kNB16;
    const bool big = use_ns16(cus, B, N, 4, 2, 2);
    double c1 = 0.5, c2 = 0.75;
#ifdef GG_AB
    if (const char *e = getenv("GG_AB_NS16_CUT")) sscanf(e, "%lf,%lf", &c1, &c2);
#endif
    if (big) {
      int grid16;
      AgeSplit as = ns16_grid(k_next_states16<19>, cus, ngroups, N, (uint32_t)(c1 * 65536.0), (uint32_t)(c2 * 65536.0), grid16);
#ifdef GG_AB
      if (const char *e = getenv("GG_AB_NS16_GRID")) { grid16 = atoi(e); as.cols = 0; }
      {
        const int dbg = getenv("GG_AB_NS16_DBG") ? atoi(getenv("GG_AB_NS16_DBG")) : 0;
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(gg::gg_ns16_dbg), &dbg, sizeof dbg, 0, hipMemcpyHostToDevice, s);
      }
#endif
      GG_DISPATCH(N, (k_next_states16<9><<<grid16, kWave, 0, s>>>(in, actions, out, status, B, canonical, as)),
                  (k_next_states16<13><<<grid16, kWave, 0, s>>>(in, actions, out, status, B, canonical, as)),
                  (k_next_states16<19><<<grid16, kWave, 0, s>>>(in, actions, out, status, B, canonical, as)));
      return (int32_t)hipGetLastError();
    }
  }
  const int64_t npairs = (B + 1) / 2;
  // up to two waves per SIMD (9x9: four) there is no pipeline to fill: one pair per wave, every read of the pair in flight at
  // once, four-wave workgroups (k_next_states2s; hipGraph node, 9x9 / 13x13 / 19x19 x 4 096 games 6.53 -> 5.88 / 7.82 -> 6.96 / 9.18 -> 8.66 us:
  // profiles/r05p_perply_nodes.txt)
  bool straight = npairs <= (int64_t)cus * (N <= 9 ? 16 : 8);
#ifdef GG_AB
  if (const char *e = getenv("GG_AB_NS_STRAIGHT")) straight = atoi(e) != 0 && npairs <= (int64_t)cus * 24;
#endif
  if (straight) {
    const unsigned grid4 = (unsigned)((npairs + 3) / 4);
    GG_DISPATCH(N, (k_next_states2s<9, 4><<<grid4, 4 * kWave, 0, s>>>(in, actions, out, status, B, N, inv, canonical)),
                (k_next_states2s<13, 4><<<grid4, 4 * kWave, 0, s>>>(in, actions, out, status, B, N, inv, canonical)),
                (k_next_states2s<19, 4><<<grid4, 4 * kWave, 0, s>>>(in, actions, out, status, B, N, inv, canonical)));
    return (int32_t)hipGetLastError();
  }
  int grid = grid_resident(cus, npairs, GG_LB_PLY);
  // from six pairs per resident wave on, the three waves of a SIMD share its pairs unevenly (k_next_states2): fractions
  // of a SIMD's pairs taken by its oldest / by its two oldest waves, 16.16 fixed point
  int cols = npairs >= (int64_t)cus * 4 * GG_LB_PLY * 2 ? cus * 4 : 0;
  uint32_t share1 = (uint32_t)(0.45 * 65536), share2 = (uint32_t)(0.79 * 65536);   // (14 / 11 / 7 of 32 pairs: 65.5 -> 61.6 us per 65 536 boards)
#ifdef GG_AB
  if (const char *e = getenv("GG_AB_NS_GRID")) { grid = atoi(e); cols = 0; }
  if (const char *e = getenv("GG_AB_NS_S1")) share1 = (uint32_t)(atof(e) * 65536);
  if (const char *e = getenv("GG_AB_NS_S2")) share2 = (uint32_t)(atof(e) * 65536);
  if (getenv("GG_AB_NS_EVEN")) cols = 0;
#endif
  if (cols) grid = cols * GG_LB_PLY;
  GG_DISPATCH(N, (k_next_states2<9><<<grid, kWave, 0, s>>>(in, actions, out, status, B, N, inv, canonical, cols, share1, share2)),
              (k_next_states2<13><<<grid, kWave, 0, s>>>(in, actions, out, status, B, N, inv, canonical, cols, share1, share2)),
              (k_next_states2<19><<<grid, kWave, 0, s>>>(in, actions, out, status, B, N, inv, canonical, cols, share1, share2)));
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_next_states_ws(const uint8_t *in, const int32_t *actions, uint8_t *out, int32_t *status, uint32_t *workspace,
                                int64_t B, int32_t N, int32_t canonical, void *hip_stream) {
  GG_ENTER(in);
  if (!actions || !out || !workspace) return GG_E_NULLPTR;
  int grid;
  const int nb = boards_per_wave(cus, B, grid);
  EnvArgs env = EnvArgs();
  env.status = status; env.states_out = out; env.ws = workspace; env.canonical = canonical;
  GG_DISPATCH4(N, 3, true, grid, const_cast<uint8_t *>(in), nullptr, nullptr, nullptr, B, N, inv, 1, 0, nb, actions, nullptr, env);
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_invalid_mask(const uint8_t *states, const int32_t *ko, uint8_t *mask, int64_t B, int32_t N,
                              void *hip_stream) {
  GG_ENTER(states);
  if (!mask) return GG_E_NULLPTR;
  {   // big batches of full-size boards: the class-major analysis, sixteen boards per wave (as gg_batch_next_states;
      // per 65 536 boards 33.3 -> 28.6 us at 13x13, 29.1 -> 18.8 us at 9x9.  At 19x19 - three waves per SIMD, one
      // workgroup per group - it was no faster than the two-board kernel at four (round 3: 47 us both: 4 096 groups on
      // 3 072 resident waves are one round and a third); with the groups of a SIMD split 2 : 1 : 1 over its three waves by
      // age it is, from four groups per SIMD on (round 4: 65 536 boards 49.0 -> 47.0 us, 131 072: 87.3 -> 81.3; 49 152:
      // 37.0 -> 38.5, so not below))
    const int64_t ngroups = (B + kNB16 - 1) / kNB16;
    const bool big = use_ns16(cus, B, N, 4, 2, 1);
    if (big) {
      int grid16;   // (19x19: three waves per SIMD share a SIMD's groups 2 : 1 : 1 by age, like gg_batch_next_states)
      const AgeSplit as = ns16_grid(k_invalid_mask16<19>, cus, ngroups, N, 32768u, 49152u, grid16);
      if (N == 9) k_invalid_mask16<9><<<grid16, kWave, 0, s>>>(states, ko, mask, B, as);
      else if (N == 13) k_invalid_mask16<13><<<grid16, kWave, 0, s>>>(states, ko, mask, B, as);
      else k_invalid_mask16<19><<<grid16, kWave, 0, s>>>(states, ko, mask, B, as);
      return (int32_t)hipGetLastError();
    }
  }
  const int64_t npairs = (B + 1) / 2;
#define GG_K(R, F) launch_pairs(k_invalid_mask2<R, F>, cus, npairs, true, s, states, ko, mask, B, N, inv)
  GG_DISPATCH_N(N);
#undef GG_K
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_areas(const uint8_t *states, int32_t *black, int32_t *white, int64_t B, int32_t N, void *hip_stream) {
  GG_ENTER(states);
  if (!black || !white) return GG_E_NULLPTR;
  const int64_t groups = (B + LdsAreas<19>::kBoards - 1) / LdsAreas<19>::kBoards;   // one wave per sixteen boards
  if (groups > 0x7FFFFFFF) return GG_E_BADSIZE;
  const int grid = (int)groups;
#define GG_K(R, F) k_areas4<R, F><<<grid, kWave, 0, s>>>(states, black, white, B, N)
  GG_DISPATCH_N(N);
#undef GG_K
  return (int32_t)hipGetLastError();
}

// children: the per-parent analysis (4 flood batches at 19x19) is repeated by every chunk of a parent's slots, so chunks
// only serve to fill the machine: enough work items for two rounds of the resident waves (4 per SIMD), one chunk per
// parent from 8 192 parents up (round 2, on a box whose stores are not the limit: 1 / 2 / 4 / 8 chunks per parent
// 6.64 / 6.31 / 6.19 / 6.02e6 parents/s)
static int children_chunks(int cus, int64_t B, int A) {
  int chunks = (int)(((int64_t)cus * 32 + B - 1) / B);
#ifdef GG_AB
  if (const char *e = getenv("GG_AB_CHUNKS")) chunks = atoi(e);
#endif
  if (chunks < 1) chunks = 1;
  if (chunks > A) chunks = A;
  return chunks;
}

int32_t gg_batch_children(const uint8_t *states, uint8_t *children, int64_t B, int32_t N, int32_t canonical,
                          void *hip_stream) {
  GG_ENTER(states);
  if (!children) return GG_E_NULLPTR;
  const int chunks = children_chunks(cus, B, N * N + 1);
  const int grid = grid_for(cus, B * chunks);
  GG_DISPATCH(N, (k_children3<9><<<grid, kWave, 0, s>>>(states, children, B, N, inv, canonical, chunks)),
              (k_children3<13><<<grid, kWave, 0, s>>>(states, children, B, N, inv, canonical, chunks)),
              (k_children3<19><<<grid, kWave, 0, s>>>(states, children, B, N, inv, canonical, chunks)));
  return (int32_t)hipGetLastError();
}

// un-padded children (gogame.children(padded=False)): per-parent counts -> exclusive offsets, then the same kernel with every
// child at the rank of its action among the kept ones
int32_t gg_batch_children_offsets(const uint8_t *states, int32_t *offsets, int32_t *order, int64_t B, int32_t N, void *hip_stream) {
  if (N >= 2 && N <= GG_MAX_BOARD && B > 0 && B > (int64_t)0x7FFFFFFF / (N * N + 1)) return GG_E_BADSIZE;
  if (B == 0 && offsets && !check(B, N)) {   // an empty batch has offsets[0] = 0
    OnDeviceOf on_dev(offsets);
    return (int32_t)hipMemsetAsync(offsets, 0, sizeof(int32_t), (hipStream_t)hip_stream);
  }
  GG_ENTER(states);
  if (!offsets) return GG_E_NULLPTR;
  k_children_counts<<<grid_for(cus, (B + 3) / 4), 4 * kWave, 0, s>>>(states, offsets, B, N);
  k_children_order_scan<<<1, 1024, 0, s>>>(offsets, order, B, N * N + 1);   // the launch order from the counts, then counts -> offsets
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_children_compact(const uint8_t *states, const int32_t *offsets, const int32_t *order, uint8_t *children, int64_t B,
                                  int32_t N, int32_t canonical, void *hip_stream) {
  GG_ENTER(states);
  if (!children || !offsets) return GG_E_NULLPTR;
  if (B > (int64_t)0x7FFFFFFF / (N * N + 1)) return GG_E_BADSIZE;
  const int chunks = children_chunks(cus, B, N * N + 1);
  const int grid = grid_for(cus, B * chunks);
  GG_DISPATCH(N, (k_children3<9, false, true><<<grid, kWave, 0, s>>>(states, children, B, N, inv, canonical, chunks, offsets, order)),
              (k_children3<13, false, true><<<grid, kWave, 0, s>>>(states, children, B, N, inv, canonical, chunks, offsets, order)),
              (k_children3<19, false, true><<<grid, kWave, 0, s>>>(states, children, B, N, inv, canonical, chunks, offsets, order)));
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_rollout(uint8_t *states, uint64_t *rng, int32_t *last_actions, int64_t *steps_done, int64_t B,
                         int32_t N, int32_t plies, int32_t auto_reset, void *hip_stream) {
  if (plies < 0) return GG_E_BADARG;
  GG_ENTER(states);
  if (plies == 0) return 0;
  if (!rng) return GG_E_NULLPTR;
  if (use_lat(cus, B, N, plies)) {   // an under-filled machine: one row per lane, the ply in registers (gg_lat.h, launched from gg_rollout.hip)
    launch_rollout_lat(0, states, rng, last_actions, steps_done, B, N, plies, auto_reset, false, s);
    return (int32_t)hipGetLastError();
  }
  if (use_rollout5(cus, B, N, plies)) {   // a full machine: 32 boards per wave, the floods of a ply as a job list (gg_v5.h)
    int grid;
    const int nb = boards_per_wave5(cus, B, grid);
    launch_rollout5(0, N, states, rng, last_actions, steps_done, B, inv, plies, auto_reset, nb, grid, s);
    return (int32_t)hipGetLastError();
  }
  if (use_multi_ply(cus, B, plies)) {   // liberty classes carried across the plies, 16 boards per wave
    int grid;
    const int nb = boards_per_wave(cus, B, grid);
    launch_rollout4(0, states, rng, last_actions, steps_done, B, N, inv, plies, auto_reset, nb, grid, s);
    return (int32_t)hipGetLastError();
  }
  if (plies == 1) {   // one ply per launch on a big batch of full-size boards: the env step without the GoEnv outputs (gg_ns16.h)
    const int64_t ngroups = (B + kNB16 - 1) / kNB16;
    const bool big = use_ns16(cus, B, N, 3, 2, 1);
    if (big) {
      int grid16;
      const AgeSplit as = ns16_grid(k_env_step16<19, false>, cus, ngroups, N, 32768u, 49152u, grid16);
      GG_DISPATCH(N, (k_env_step16<9, false><<<grid16, kWave, 0, s>>>(states, nullptr, rng, nullptr, nullptr, nullptr, nullptr, B, 0.f, auto_reset, as, last_actions, steps_done)),
                  (k_env_step16<13, false><<<grid16, kWave, 0, s>>>(states, nullptr, rng, nullptr, nullptr, nullptr, nullptr, B, 0.f, auto_reset, as, last_actions, steps_done)),
                  (k_env_step16<19, false><<<grid16, kWave, 0, s>>>(states, nullptr, rng, nullptr, nullptr, nullptr, nullptr, B, 0.f, auto_reset, as, last_actions, steps_done)));
      return (int32_t)hipGetLastError();
    }
  }
  const int64_t npairs = (B + 1) / 2;
  if (plies <= 2) {
    // small launches go out as four-wave workgroups: the dispatcher's ramp is per workgroup (k_rollout2, WPB).  hipGraph node,
    // one ply, single-wave -> four-wave workgroups (profiles/r05p_wpb.txt): 9x9 4 096 / 8 192 games 6.02 -> 5.82 / 7.97 -> 7.61 us,
    // 13x13 4 096 6.98 -> 6.80 (8 192: 9.67 -> 10.66, the four waves' LDS no longer fits three times per CU), 19x19 4 096 8.49 ->
    // 8.37; 1 024 games: no change -> up to two waves per SIMD (9x9: four)
    int wpb = npairs <= (int64_t)cus * (N <= 9 ? 16 : 8) ? 4 : 1;
#ifdef GG_AB
    if (const char *e = getenv("GG_AB_WPB")) wpb = atoi(e) == 4 && npairs < (int64_t)cus * 24 ? 4 : 1;
#endif
    if (wpb == 4) {
      const AgeSplit none = {0, {0, 0, 0}};
      const unsigned grid4 = (unsigned)((npairs + 3) / 4);
#define GG_K(R, F) k_rollout2_w4<R, F><<<grid4, 4 * kWave, 0, s>>>(states, rng, last_actions, steps_done, B, N, inv, plies, auto_reset, none)
      GG_DISPATCH_N(N);
#undef GG_K
      return (int32_t)hipGetLastError();
    }
#define GG_K(R, F) launch_pairs(k_rollout2<R, true, false, F>, cus, npairs, true, s, states, rng, last_actions, steps_done, B, N, inv, plies, auto_reset)
    GG_DISPATCH_N(N);
#undef GG_K
  } else {
#define GG_K(R, F) launch_pairs(k_rollout2<R, false, false, F>, cus, npairs, plies < 8, s, states, rng, last_actions, steps_done, B, N, inv, plies, auto_reset)
    GG_DISPATCH_N(N);
#undef GG_K
  }
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_env_step(uint8_t *states, const int32_t *actions, uint64_t *rng, float *rewards, uint8_t *dones,
                          int32_t *status, int32_t *taken_actions, int64_t B, int32_t N, float komi,
                          int32_t reward_method, int32_t auto_reset, void *hip_stream) {
  if (reward_method != GG_REWARD_REAL && reward_method != GG_REWARD_HEURISTIC) return GG_E_BADARG;
  GG_ENTER(states);
  if (!actions && !rng) return GG_E_NULLPTR;
  {   // big batches of full-size boards: the class-major analysis, sixteen boards per wave (as gg_batch_next_states)
    const int64_t ngroups = (B + kNB16 - 1) / kNB16;
    const bool big = use_ns16(cus, B, N, 3, 2, 1);
    if (big) {
      int grid16;   // 19x19: a SIMD's groups 2 : 1 : 1 by wave age
      const AgeSplit as = reward_method == GG_REWARD_HEURISTIC ? ns16_grid(k_env_step16<19, true>, cus, ngroups, N, 32768u, 49152u, grid16)
                                                               : ns16_grid(k_env_step16<19, false>, cus, ngroups, N, 32768u, 49152u, grid16);
      if (reward_method == GG_REWARD_HEURISTIC) {
        GG_DISPATCH(N, (k_env_step16<9, true><<<grid16, kWave, 0, s>>>(states, actions, rng, rewards, dones, status, taken_actions, B, komi, auto_reset, as)),
                    (k_env_step16<13, true><<<grid16, kWave, 0, s>>>(states, actions, rng, rewards, dones, status, taken_actions, B, komi, auto_reset, as)),
                    (k_env_step16<19, true><<<grid16, kWave, 0, s>>>(states, actions, rng, rewards, dones, status, taken_actions, B, komi, auto_reset, as)));
      } else {
        GG_DISPATCH(N, (k_env_step16<9, false><<<grid16, kWave, 0, s>>>(states, actions, rng, rewards, dones, status, taken_actions, B, komi, auto_reset, as)),
                    (k_env_step16<13, false><<<grid16, kWave, 0, s>>>(states, actions, rng, rewards, dones, status, taken_actions, B, komi, auto_reset, as)),
                    (k_env_step16<19, false><<<grid16, kWave, 0, s>>>(states, actions, rng, rewards, dones, status, taken_actions, B, komi, auto_reset, as)));
      }
      return (int32_t)hipGetLastError();
    }
  }
  const int64_t npairs = (B + 1) / 2;
  bool w4 = npairs <= (int64_t)cus * (N <= 9 ? 16 : 8);   // small launches: four-wave workgroups (gg_batch_rollout, k_rollout2_w4)
#ifdef GG_AB
  if (const char *e = getenv("GG_AB_WPB")) w4 = atoi(e) == 4 && npairs < (int64_t)cus * 24;
#endif
  if (w4) {
    const AgeSplit none = {0, {0, 0, 0}};
    const unsigned grid4 = (unsigned)((npairs + 3) / 4);
    if (reward_method == GG_REWARD_HEURISTIC) {
#define GG_K(R, F) k_env_step2_w4<R, true, F><<<grid4, 4 * kWave, 0, s>>>(states, actions, rng, rewards, dones, status, taken_actions, B, N, inv, komi, auto_reset, none)
      GG_DISPATCH_N(N);
#undef GG_K
    } else {
#define GG_K(R, F) k_env_step2_w4<R, false, F><<<grid4, 4 * kWave, 0, s>>>(states, actions, rng, rewards, dones, status, taken_actions, B, N, inv, komi, auto_reset, none)
      GG_DISPATCH_N(N);
#undef GG_K
    }
    return (int32_t)hipGetLastError();
  }
  if (reward_method == GG_REWARD_HEURISTIC) {
#define GG_K(R, F) launch_pairs(k_env_step2<R, true, false, F>, cus, npairs, true, s, states, actions, rng, rewards, dones, status, taken_actions, B, N, inv, komi, auto_reset)
    GG_DISPATCH_N(N);
#undef GG_K
  } else {
#define GG_K(R, F) launch_pairs(k_env_step2<R, false, false, F>, cus, npairs, true, s, states, actions, rng, rewards, dones, status, taken_actions, B, N, inv, komi, auto_reset)
    GG_DISPATCH_N(N);
#undef GG_K
  }
  return (int32_t)hipGetLastError();
}

// gg_batch_env_step + the Tromp-Taylor areas of every resulting position, ONE launch: the instantiation that scores every
// game every step (the two area floods ride in idle flood lanes of the liberty analysis) with the reward formula of the
// caller's choice.  GoEnv.step (gymgo_amd/envs/go_env.py) is this launch at B = 1.
int32_t gg_batch_env_step_scored(uint8_t *states, const int32_t *actions, uint64_t *rng, float *rewards, uint8_t *dones,
                                 int32_t *status, int32_t *taken_actions, int32_t *areas, int64_t B, int32_t N, float komi,
                                 int32_t reward_method, int32_t auto_reset, void *hip_stream) {
  if (reward_method != GG_REWARD_REAL && reward_method != GG_REWARD_HEURISTIC) return GG_E_BADARG;
  GG_ENTER(states);
  if (!actions && !rng) return GG_E_NULLPTR;
  if (!areas) return GG_E_NULLPTR;
  const int real = reward_method == GG_REWARD_REAL ? 1 : 0;
  const int64_t npairs = (B + 1) / 2;
  if (npairs <= (int64_t)cus * (N <= 9 ? 16 : 8)) {   // small launches: four-wave workgroups (gg_batch_env_step)
    const AgeSplit none = {0, {0, 0, 0}};
    const unsigned grid4 = (unsigned)((npairs + 3) / 4);
#define GG_K(R, F) k_env_step2_w4<R, true, F><<<grid4, 4 * kWave, 0, s>>>(states, actions, rng, rewards, dones, status, taken_actions, B, N, inv, komi, auto_reset, none, areas, real)
    GG_DISPATCH_N(N);
#undef GG_K
    return (int32_t)hipGetLastError();
  }
#define GG_K(R, F)                                                                                                          \
  do {                                                                                                                      \
    int grid;                                                                                                               \
    const AgeSplit as = age_split(reinterpret_cast<const void *>(k_env_step2<R, true, false, F>), cus, npairs, true, grid); \
    k_env_step2<R, true, false, F><<<grid, kWave, 0, s>>>(states, actions, rng, rewards, dones, status, taken_actions, B, N, inv, komi, auto_reset, as, areas, real); \
  } while (0)
  GG_DISPATCH_N(N);
#undef GG_K
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_sample_actions(const uint8_t *states, uint64_t *rng, int32_t *actions, int64_t B, int32_t N,
                                void *hip_stream) {
  GG_ENTER(states);
  if (!rng || !actions) return GG_E_NULLPTR;
  const int64_t groups = (B + LdsSample<19>::kBoards - 1) / LdsSample<19>::kBoards;   // one wave per sixteen boards
  if (groups > 0x7FFFFFFF) return GG_E_BADSIZE;
  const int grid = (int)groups;
  GG_DISPATCH(N, (k_sample16<9><<<grid, kWave, 0, s>>>(states, rng, actions, B, N)),
              (k_sample16<13><<<grid, kWave, 0, s>>>(states, rng, actions, B, N)),
              (k_sample16<19><<<grid, kWave, 0, s>>>(states, rng, actions, B, N)));
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_update_pieces(uint8_t *states, const int32_t *adj, int32_t K, const int32_t *players, uint8_t *killed,
                               int64_t B, int32_t N, void *hip_stream) {
  if (K < 0) return GG_E_BADARG;
  GG_ENTER(states);
  if (!players || (K > 0 && !adj)) return GG_E_NULLPTR;
  const int grid = grid_for(cus, B);
  GG_DISPATCH(N, (k_update_pieces<9><<<grid, kWave, 0, s>>>(states, adj, K, players, killed, B, N, inv)),
              (k_update_pieces<13><<<grid, kWave, 0, s>>>(states, adj, K, players, killed, B, N, inv)),
              (k_update_pieces<19><<<grid, kWave, 0, s>>>(states, adj, K, players, killed, B, N, inv)));
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_reset_finished(uint8_t *states, int64_t B, int32_t N, void *hip_stream) {
  GG_ENTER(states);
  int64_t waves = (B + kWave - 1) / kWave;
  int blocks = (int)((waves + 3) / 4);
  if (blocks > 4096) blocks = 4096;
  k_reset_finished<<<blocks, 4 * kWave, 0, s>>>(states, B, N);
  return (int32_t)hipGetLastError();
}

int32_t gg_packed_words(int32_t N) { return (N < 2 || N > GG_MAX_BOARD) ? GG_E_BADSIZE : 3 * N + 1; }

int32_t gg_batch_pack_states(const uint8_t *states, uint32_t *packed, int64_t B, int32_t N, void *hip_stream) {
  GG_ENTER(states);
  if (!packed) return GG_E_NULLPTR;
  const int grid = grid_for(cus, (B + 1) / 2);
  GG_DISPATCH(N, (k_pack<9><<<grid, kWave, 0, s>>>(states, packed, B, N)),
              (k_pack<13><<<grid, kWave, 0, s>>>(states, packed, B, N)),
              (k_pack<19><<<grid, kWave, 0, s>>>(states, packed, B, N)));
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_unpack_states(const uint32_t *packed, uint8_t *states, int64_t B, int32_t N, void *hip_stream) {
  GG_ENTER(packed);
  if (!states) return GG_E_NULLPTR;
  const int grid = grid_for(cus, (B + 1) / 2);
  GG_DISPATCH(N, (k_unpack<9><<<grid, kWave, 0, s>>>(packed, states, B, N, 3)),
              (k_unpack<13><<<grid, kWave, 0, s>>>(packed, states, B, N, 3)),
              (k_unpack<19><<<grid, kWave, 0, s>>>(packed, states, B, N, 3)));
  return (int32_t)hipGetLastError();
}

// ---- the same operations on packed boards (uint32 [B][3 N + 1], see gg_batch_pack_states): no byte-plane conversions
int32_t gg_batch_next_states_packed(const uint32_t *in, const int32_t *actions, uint32_t *out, int32_t *status, int64_t B,
                                    int32_t N, int32_t canonical, void *hip_stream) {
  GG_ENTER(in);
  if (!actions || !out) return GG_E_NULLPTR;
  const int64_t npairs = (B + 1) / 2;
#define GG_K(R, F) launch_pairs(k_next_states_p<R, F>, cus, npairs, true, s, in, actions, out, status, B, N, inv, canonical)
  GG_DISPATCH_N(N);
#undef GG_K
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_rollout_packed(uint32_t *packed, uint64_t *rng, int32_t *last_actions, int64_t *steps_done, int64_t B,
                                int32_t N, int32_t plies, int32_t auto_reset, void *hip_stream) {
  if (plies < 0) return GG_E_BADARG;
  GG_ENTER(packed);
  if (plies == 0) return 0;
  if (!rng) return GG_E_NULLPTR;
  uint8_t *st = reinterpret_cast<uint8_t *>(packed);
  if (use_multi_ply(cus, B, plies)) {
    int grid3;
    const int nb = boards_per_wave(cus, B, grid3);
    launch_rollout4(1, st, rng, last_actions, steps_done, B, N, inv, plies, auto_reset, nb, grid3, s);
    return (int32_t)hipGetLastError();
  }
  const int64_t npairs = (B + 1) / 2;
#define GG_K(R, F) launch_pairs(k_rollout2<R, false, true, F>, cus, npairs, plies < 8, s, st, rng, last_actions, steps_done, B, N, inv, plies, auto_reset)
  GG_DISPATCH_N(N);
#undef GG_K
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_env_step_packed(uint32_t *packed, const int32_t *actions, uint64_t *rng, float *rewards, uint8_t *dones,
                                 int32_t *status, int32_t *taken_actions, int64_t B, int32_t N, float komi,
                                 int32_t reward_method, int32_t auto_reset, void *hip_stream) {
  if (reward_method != GG_REWARD_REAL && reward_method != GG_REWARD_HEURISTIC) return GG_E_BADARG;
  GG_ENTER(packed);
  if (!actions && !rng) return GG_E_NULLPTR;
  const int64_t npairs = (B + 1) / 2;
  uint8_t *st = reinterpret_cast<uint8_t *>(packed);
  if (reward_method == GG_REWARD_HEURISTIC) {
#define GG_K(R, F) launch_pairs(k_env_step2<R, true, true, F>, cus, npairs, true, s, st, actions, rng, rewards, dones, status, taken_actions, B, N, inv, komi, auto_reset)
    GG_DISPATCH_N(N);
#undef GG_K
  } else {
#define GG_K(R, F) launch_pairs(k_env_step2<R, false, true, F>, cus, npairs, true, s, st, actions, rng, rewards, dones, status, taken_actions, B, N, inv, komi, auto_reset)
    GG_DISPATCH_N(N);
#undef GG_K
  }
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_children_packed(const uint32_t *packed, uint32_t *children, int64_t B, int32_t N, int32_t canonical,
                                 void *hip_stream) {
  GG_ENTER(packed);
  if (!children) return GG_E_NULLPTR;
  const int chunks = children_chunks(cus, B, N * N + 1);
  const int grid = grid_for(cus, B * chunks);
  const uint8_t *st = reinterpret_cast<const uint8_t *>(packed);
  uint8_t *ch = reinterpret_cast<uint8_t *>(children);
  GG_DISPATCH(N, (k_children3<9, true><<<grid, kWave, 0, s>>>(st, ch, B, N, inv, canonical, chunks)),
              (k_children3<13, true><<<grid, kWave, 0, s>>>(st, ch, B, N, inv, canonical, chunks)),
              (k_children3<19, true><<<grid, kWave, 0, s>>>(st, ch, B, N, inv, canonical, chunks)));
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_play_moves(uint8_t *states, const int32_t *moves, int32_t *played, int64_t B, int32_t N, int32_t T,
                            void *hip_stream) {
  if (T < 0) return GG_E_BADARG;
  GG_ENTER(states);
  if (T > 0 && !moves) return GG_E_NULLPTR;
  if (use_multi_ply(cus, B, T)) {
    int grid3;
    const int nb = boards_per_wave(cus, B, grid3);
    GG_DISPATCH4(N, 0, true, grid3, states, nullptr, nullptr, nullptr, B, N, inv, T, 0, nb, moves, played);
    return (int32_t)hipGetLastError();
  }
  const int64_t npairs = (B + 1) / 2;
  GG_DISPATCH(N, (launch_pairs(k_play_moves2<9, false>, cus, npairs, true, s, states, moves, played, B, N, inv, T)),
              (launch_pairs(k_play_moves2<13, false>, cus, npairs, true, s, states, moves, played, B, N, inv, T)),
              (launch_pairs(k_play_moves2<19, false>, cus, npairs, true, s, states, moves, played, B, N, inv, T)));
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_play_moves_packed(uint32_t *packed, const int32_t *moves, int32_t *played, int64_t B, int32_t N, int32_t T,
                                   void *hip_stream) {
  if (T < 0) return GG_E_BADARG;
  GG_ENTER(packed);
  if (T > 0 && !moves) return GG_E_NULLPTR;
  uint8_t *st = reinterpret_cast<uint8_t *>(packed);
  if (use_multi_ply(cus, B, T)) {
    int grid3;
    const int nb = boards_per_wave(cus, B, grid3);
    GG_DISPATCH4(N, 1, true, grid3, st, nullptr, nullptr, nullptr, B, N, inv, T, 0, nb, moves, played);
    return (int32_t)hipGetLastError();
  }
  const int64_t npairs = (B + 1) / 2;
  GG_DISPATCH(N, (launch_pairs(k_play_moves2<9, true>, cus, npairs, true, s, st, moves, played, B, N, inv, T)),
              (launch_pairs(k_play_moves2<13, true>, cus, npairs, true, s, st, moves, played, B, N, inv, T)),
              (launch_pairs(k_play_moves2<19, true>, cus, npairs, true, s, st, moves, played, B, N, inv, T)));
  return (int32_t)hipGetLastError();
}

// ---- tracked boards (uint32 [B][5 N + 1]): packed boards that carry their liberty classes (see gg_v4.h)
int32_t gg_tracked_words(int32_t N) { return (N < 2 || N > GG_MAX_BOARD) ? GG_E_BADSIZE : 5 * N + 1; }

int32_t gg_batch_track_states(const uint8_t *states, uint32_t *tracked, int64_t B, int32_t N, void *hip_stream) {
  GG_ENTER(states);
  if (!tracked) return GG_E_NULLPTR;
  {   // big batches of full-size boards: the class-major analysis, sixteen boards per wave (as gg_batch_invalid_mask)
    const int64_t ngroups = (B + kNB16 - 1) / kNB16;
    // (us per 65 536 / 131 072 boards, two-board kernel -> sixteen-board kernel: 19x19 48.3 -> 47.3 / 110.5 -> 102.1,
    // 13x13 39.0 -> 30.7 / 80.4 -> 62.7, 9x9 30.3 -> 22.0 / 52.6 -> 35.4; the take-over sizes are the mask's)
    if (use_ns16(cus, B, N, 4, 2, 1)) {
      int grid16;
      const AgeSplit as = ns16_grid(k_track16<19>, cus, ngroups, N, 32768u, 49152u, grid16);
      if (N == 9) k_track16<9><<<grid16, kWave, 0, s>>>(states, tracked, B, as);
      else if (N == 13) k_track16<13><<<grid16, kWave, 0, s>>>(states, tracked, B, as);
      else k_track16<19><<<grid16, kWave, 0, s>>>(states, tracked, B, as);
      return (int32_t)hipGetLastError();
    }
  }
  const int64_t npairs = (B + 1) / 2;
  GG_DISPATCH(N, (launch_pairs(k_track<9>, cus, npairs, true, s, states, tracked, B, N, inv)),
              (launch_pairs(k_track<13>, cus, npairs, true, s, states, tracked, B, N, inv)),
              (launch_pairs(k_track<19>, cus, npairs, true, s, states, tracked, B, N, inv)));
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_untrack_states(const uint32_t *tracked, uint8_t *states, int64_t B, int32_t N, void *hip_stream) {
  GG_ENTER(tracked);
  if (!states) return GG_E_NULLPTR;
  const int grid = grid_for(cus, (B + 1) / 2);
  GG_DISPATCH(N, (k_unpack<9><<<grid, kWave, 0, s>>>(tracked, states, B, N, 5)),
              (k_unpack<13><<<grid, kWave, 0, s>>>(tracked, states, B, N, 5)),
              (k_unpack<19><<<grid, kWave, 0, s>>>(tracked, states, B, N, 5)));
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_rollout_tracked(uint32_t *tracked, uint64_t *rng, int32_t *last_actions, int64_t *steps_done, int64_t B,
                                 int32_t N, int32_t plies, int32_t auto_reset, void *hip_stream) {
  if (plies < 0) return GG_E_BADARG;
  GG_ENTER(tracked);
  if (plies == 0) return 0;
  if (!rng) return GG_E_NULLPTR;
  uint8_t *st = reinterpret_cast<uint8_t *>(tracked);
  if (use_lat(cus, B, N, plies, true)) {   // an under-filled machine: one row per lane, the ply in registers (gg_lat.h)
    launch_rollout_lat(2, st, rng, last_actions, steps_done, B, N, plies, auto_reset, lat_w4(cus, B, N, plies), s);
    return (int32_t)hipGetLastError();
  }
  if (use_rollout5(cus, B, N, plies)) {
    int grid5;
    const int nb5 = boards_per_wave5(cus, B, grid5);
    launch_rollout5(2, N, st, rng, last_actions, steps_done, B, inv, plies, auto_reset, nb5, grid5, s);
    return (int32_t)hipGetLastError();
  }
  int grid3;
  const int nb = boards_per_wave(cus, B, grid3);
  launch_rollout4(2, st, rng, last_actions, steps_done, B, N, inv, plies, auto_reset, nb, grid3, s);
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_play_moves_tracked(uint32_t *tracked, const int32_t *moves, int32_t *played, int64_t B, int32_t N, int32_t T,
                                    void *hip_stream) {
  if (T < 0) return GG_E_BADARG;
  GG_ENTER(tracked);
  if (T > 0 && !moves) return GG_E_NULLPTR;
  uint8_t *st = reinterpret_cast<uint8_t *>(tracked);
  int grid3;
  const int nb = boards_per_wave(cus, B, grid3);
  GG_DISPATCH4(N, 2, true, grid3, st, nullptr, nullptr, nullptr, B, N, inv, T, 0, nb, moves, played);
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_env_step_tracked(uint32_t *tracked, const int32_t *actions, uint64_t *rng, float *rewards, uint8_t *dones,
                                  int32_t *status, int32_t *taken_actions, uint8_t *states_out, int64_t *steps_done,
                                  int64_t B, int32_t N, float komi, int32_t reward_method, int32_t auto_reset,
                                  void *hip_stream) {
  if (reward_method != GG_REWARD_REAL && reward_method != GG_REWARD_HEURISTIC) return GG_E_BADARG;
  GG_ENTER(tracked);
  if (!actions && !rng) return GG_E_NULLPTR;
  uint8_t *st = reinterpret_cast<uint8_t *>(tracked);
  int grid;
  const int nb = boards_per_wave(cus, B, grid);
  EnvArgs env;
  env.actions = actions; env.rewards = rewards; env.dones = dones; env.status = status; env.taken = taken_actions;
  env.states_out = states_out; env.komi = komi; env.heuristic = reward_method == GG_REWARD_HEURISTIC;
  env.ws = nullptr; env.canonical = 0; env.weights = nullptr;
  if (use_lat_env(cus, B, N, states_out != nullptr)) {   // one row per lane, the ply in registers (gg_lat.h)
    launch_env_step_lat(tracked, rng, steps_done, B, N, auto_reset, env, lat_w4(cus, B, N, 1), s);
    return (int32_t)hipGetLastError();
  }
  if (actions) {
    GG_DISPATCH4E(N, true, grid, st, nullptr, nullptr, steps_done, B, N, inv, 1, auto_reset, nb, actions, nullptr, env);
  } else {
    GG_DISPATCH4E(N, false, grid, st, rng, nullptr, steps_done, B, N, inv, 1, auto_reset, nb, nullptr, nullptr, env);
  }
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_env_step_tracked_weighted(uint32_t *tracked, const void *weights, int32_t weight_dtype, uint64_t *rng, float *rewards,
                                           uint8_t *dones, int32_t *status, int32_t *taken_actions, uint8_t *states_out,
                                           int64_t *steps_done, int64_t B, int32_t N, float komi, int32_t reward_method,
                                           int32_t auto_reset, void *hip_stream) {
  if (reward_method != GG_REWARD_REAL && reward_method != GG_REWARD_HEURISTIC) return GG_E_BADARG;
  if (weight_dtype < GG_W_F32 || weight_dtype > GG_W_F16) return GG_E_BADARG;
  GG_ENTER(tracked);
  if (!weights || !rng) return GG_E_NULLPTR;
  uint8_t *st = reinterpret_cast<uint8_t *>(tracked);
  int grid;
  const int nb = boards_per_wave(cus, B, grid);
  EnvArgs env = EnvArgs();
  env.rewards = rewards; env.dones = dones; env.status = status; env.taken = taken_actions;
  env.states_out = states_out; env.komi = komi; env.heuristic = reward_method == GG_REWARD_HEURISTIC;
  env.weights = weights; env.wdtype = weight_dtype;
  // the given-moves instantiation, with the move of every game drawn from its weights by the kernel itself
  GG_DISPATCH4W(N, grid, st, rng, nullptr, steps_done, B, N, inv, 1, auto_reset, nb, nullptr, nullptr, env);
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_sample_weighted(const uint8_t *states, const void *weights, int32_t weight_dtype, uint64_t *rng,
                                 int32_t *actions, int64_t B, int32_t N, void *hip_stream) {
  if (weight_dtype < GG_W_F32 || weight_dtype > GG_W_F16) return GG_E_BADARG;
  GG_ENTER(weights);
  if (!rng || !actions) return GG_E_NULLPTR;   // states may be NULL: nothing is masked
  const int grid = grid_for(cus, (B + 3) / 4);
  GG_DISPATCH(N, (k_sample_weighted<9><<<grid, kWave, 0, s>>>(states, weights, weight_dtype, rng, actions, B, N)),
              (k_sample_weighted<13><<<grid, kWave, 0, s>>>(states, weights, weight_dtype, rng, actions, B, N)),
              (k_sample_weighted<19><<<grid, kWave, 0, s>>>(states, weights, weight_dtype, rng, actions, B, N)));
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_sample_weighted_rows(const uint32_t *boards, int32_t planes, const void *weights, int32_t weight_dtype,
                                      uint64_t *rng, int32_t *actions, int64_t B, int32_t N, void *hip_stream) {
  if (planes != 3 && planes != 5) return GG_E_BADARG;
  if (weight_dtype < GG_W_F32 || weight_dtype > GG_W_F16) return GG_E_BADARG;
  GG_ENTER(boards);
  if (!weights || !rng || !actions) return GG_E_NULLPTR;
  const int grid = grid_for(cus, (B + 3) / 4);
  const int W = planes * N + 1;
  GG_DISPATCH(N, (k_sample_weighted_rows<9><<<grid, kWave, 0, s>>>(boards, W, weights, weight_dtype, rng, actions, B, N)),
              (k_sample_weighted_rows<13><<<grid, kWave, 0, s>>>(boards, W, weights, weight_dtype, rng, actions, B, N)),
              (k_sample_weighted_rows<19><<<grid, kWave, 0, s>>>(boards, W, weights, weight_dtype, rng, actions, B, N)));
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_symmetry(const uint8_t *in, const int32_t *orient, uint8_t *out, int64_t B, int32_t C, int32_t N,
                          void *hip_stream) {
  if (C < 1 || (int64_t)C * N * N > 8192) return GG_E_BADARG;
  GG_ENTER(in);
  if (!out) return GG_E_NULLPTR;
  const int grid = grid_for(cus, B);
  if (C * N * N <= 2304 && C * N <= 128) {   // boards of state size: through the bit domain when the planes are 0 / 1 (gg_sym.h)
    k_symmetry_bits<2304><<<grid, kWave, 0, s>>>(in, orient, out, B, C, N);
    return (int32_t)hipGetLastError();
  }
  if (C * N * N <= 2304) k_symmetry_bytes<2304><<<grid, kWave, 0, s>>>(in, orient, out, B, C, N);
  else k_symmetry_bytes<8192><<<grid, kWave, 0, s>>>(in, orient, out, B, C, N);
  return (int32_t)hipGetLastError();
}

int32_t gg_batch_symmetry_rows(const uint32_t *in, int32_t planes, const int32_t *orient, uint32_t *out, int64_t B,
                               int32_t N, void *hip_stream) {
  if (planes != 3 && planes != 5) return GG_E_BADARG;
  GG_ENTER(in);
  if (!out) return GG_E_NULLPTR;
  const int grid = grid_for(cus, (B + 1) / 2, 128);   // short iterations: 54 -> 47 us per 65 536 tracked boards with 4x the workgroups
  k_symmetry_rows<<<grid, kWave, 0, s>>>(in, orient, out, B, N, planes);
  return (int32_t)hipGetLastError();
}

int32_t gg_rng_seed(uint64_t *rng, uint64_t base_seed, int64_t first_game, int64_t B, void *hip_stream) {
  if (B < 0) return GG_E_BADSIZE;
  if (B == 0) return 0;
  if (!rng) return GG_E_NULLPTR;
  OnDeviceOf on_dev(rng);
  hipStream_t s = (hipStream_t)hip_stream;
  k_rng_seed<<<(unsigned)((B + 255) / 256), 256, 0, s>>>(rng, base_seed, first_game, B);
  return (int32_t)hipGetLastError();
}

}  // extern "C"
