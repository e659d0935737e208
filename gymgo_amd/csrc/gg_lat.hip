// gg_lat.hip - the launches of the latency-shaped multi-ply kernel (gg_lat.h: gg_batch_rollout on batches that leave the SIMDs
// under-filled) as a translation unit of their own.  Unlike gg_rollout.hip it is compiled with the DEFAULT code-generation
// switches: the post-register-allocation scheduler that costs k_rollout4 1 % gains this kernel 7 % (A/B on one box, 4 096
// games of 9x9 x 256 plies: 0.2320 ms per launch here against 0.2491 ms inside gg_rollout.hip, identical states) - its ply
// is many short dependency chains the source does not interleave.  A unit of its own also keeps the kernel's machine code
// (and the hash bench.py ties config 2's PMC record to) independent of edits elsewhere.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gg_common.h"
#include "gg_v2.h"
#include "gg_lat.h"

namespace gg {

// gg_batch_rollout on a batch that leaves the SIMDs under-filled (gg_kernels.hip: use_lat): one single-wave workgroup per
// four 9x9 / 13x13 boards or two 19x19 boards.
// io: 0 byte planes, 2 tracked boards (`st` is the batch in that format)
#define GG_LAT(R, F)                                                                                                            \
  do {                                                                                                                          \
    const unsigned grid_ = (unsigned)((B + Lat<R>::NBW - 1) / Lat<R>::NBW);                                                     \
    const unsigned grid4_ = (grid_ + 3u) / 4u;                                                                                  \
    if (io == 0) {                                                                                                              \
      if (auto_reset) k_rollout_lat<R, F, true, 0><<<grid_, kWave, 0, s>>>(st, rng, last_actions, steps_done, B, N, plies, 1);  \
      else k_rollout_lat<R, F, false, 0><<<grid_, kWave, 0, s>>>(st, rng, last_actions, steps_done, B, N, plies, 0);            \
    } else if (w4 && plies <= 2) {                                                                                              \
      if (auto_reset) k_rollout_lat_w4<R, F, true, true><<<grid4_, 4 * kWave, 0, s>>>(st, rng, last_actions, steps_done, B, N, plies, 1); \
      else k_rollout_lat_w4<R, F, false, true><<<grid4_, 4 * kWave, 0, s>>>(st, rng, last_actions, steps_done, B, N, plies, 0); \
    } else if (w4) {                                                                                                            \
      if (auto_reset) k_rollout_lat_w4<R, F, true, false><<<grid4_, 4 * kWave, 0, s>>>(st, rng, last_actions, steps_done, B, N, plies, 1); \
      else k_rollout_lat_w4<R, F, false, false><<<grid4_, 4 * kWave, 0, s>>>(st, rng, last_actions, steps_done, B, N, plies, 0); \
    } else if (plies <= 2) {                                                                                                    \
      if (auto_reset) k_rollout_lat<R, F, true, 2, true><<<grid_, kWave, 0, s>>>(st, rng, last_actions, steps_done, B, N, plies, 1); \
      else k_rollout_lat<R, F, false, 2, true><<<grid_, kWave, 0, s>>>(st, rng, last_actions, steps_done, B, N, plies, 0);      \
    } else {                                                                                                                    \
      if (auto_reset) k_rollout_lat<R, F, true, 2><<<grid_, kWave, 0, s>>>(st, rng, last_actions, steps_done, B, N, plies, 1);  \
      else k_rollout_lat<R, F, false, 2><<<grid_, kWave, 0, s>>>(st, rng, last_actions, steps_done, B, N, plies, 0);            \
    }                                                                                                                           \
  } while (0)
// w4 (tracked boards only): four waves per workgroup (k_rollout_lat_w4: short launches of few workgroups)
void launch_rollout_lat(int io, uint8_t *st, uint64_t *rng, int32_t *last_actions, int64_t *steps_done, int64_t B, int32_t N, int plies,
                        int auto_reset, bool w4, hipStream_t s) {
  if (N == 9) GG_LAT(9, true);
  else if (N < 9) GG_LAT(9, false);
  else if (N == 13) GG_LAT(13, true);
  else if (N < 13) GG_LAT(13, false);
  else if (N == 19) GG_LAT(19, true);
  else GG_LAT(19, false);
}
#undef GG_LAT

// gg_batch_env_step_tracked on a batch that leaves the SIMDs under-filled: the same one-ply kernel with GoEnv.step's outputs
#define GG_LATE(R, F)                                                                                                          \
  do {                                                                                                                         \
    const unsigned grid_ = (unsigned)((B + Lat<R>::NBW - 1) / Lat<R>::NBW);                                                    \
    const unsigned grid4_ = (grid_ + 3u) / 4u;                                                                                 \
    if (w4) {                                                                                                                  \
      if (env.actions) k_env_step_lat_w4<R, F, true><<<grid4_, 4 * kWave, 0, s>>>(tracked, rng, steps_done, B, N, auto_reset, env); \
      else k_env_step_lat_w4<R, F, false><<<grid4_, 4 * kWave, 0, s>>>(tracked, rng, steps_done, B, N, auto_reset, env);       \
    } else if (env.actions) k_env_step_lat<R, F, true><<<grid_, kWave, 0, s>>>(tracked, rng, steps_done, B, N, auto_reset, env); \
    else k_env_step_lat<R, F, false><<<grid_, kWave, 0, s>>>(tracked, rng, steps_done, B, N, auto_reset, env);                 \
  } while (0)
void launch_env_step_lat(uint32_t *tracked, uint64_t *rng, int64_t *steps_done, int64_t B, int32_t N, int auto_reset,
                         const EnvArgs &env, bool w4, hipStream_t s) {
  if (N == 9) GG_LATE(9, true);
  else if (N < 9) GG_LATE(9, false);
  else if (N == 13) GG_LATE(13, true);
  else if (N < 13) GG_LATE(13, false);
  else if (N == 19) GG_LATE(19, true);
  else GG_LATE(19, false);
}
#undef GG_LATE

}  // namespace gg

#ifdef GG_AB_PROF
// A/B builds only: read and clear the phase clocks of THIS translation unit's launches (gg_prof has internal linkage)
GG_PROF_READ(gg_ab_prof_read_lat)
GG_PROF_RAW(gg_ab_prof_raw_lat)
#endif
