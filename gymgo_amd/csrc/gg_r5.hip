// gg_r5.hip - the launches of the thirty-two-board multi-ply kernel (gg_v5.h: k_rollout5) as a translation unit of their own,
// compiled with the DEFAULT code-generation switches: unlike k_rollout4 (gg_rollout.hip, built without the post-RA machine
// scheduler) this kernel runs two waves per SIMD and lives on the instruction-level parallelism of its ten rows per lane, which
// the post-RA scheduler interleaves: 1.579 -> 1.560 ms per launch of 65 536 games x 256 plies (A/B on one lease, identical states).
// Argument checks, device selection and grid sizing stay in gg_kernels.hip, which calls launch_rollout5().
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gg_common.h"
#include "gg_v2.h"
#include "gg_v4.h"
#include "gg_v5.h"
#include "gg_ws.h"   // (defines the weighted-draw helpers gg_v4.h names)

namespace gg {

// full-size 19x19 boards, byte planes (io 0) or tracked boards (io 2)
void launch_rollout5(int io, int N, uint8_t *st, uint64_t *rng, int32_t *last_actions, int64_t *steps_done, int64_t B, uint32_t inv,
                     int plies, int auto_reset, int nb, int grid, hipStream_t s) {
#define GG_R5(R)                                                                                                              \
  do {                                                                                                                        \
    if (io == 0) k_rollout5<R, 0><<<grid, kWave, 0, s>>>(st, rng, last_actions, steps_done, B, inv, plies, auto_reset, nb);   \
    else k_rollout5<R, 2><<<grid, kWave, 0, s>>>(st, rng, last_actions, steps_done, B, inv, plies, auto_reset, nb);           \
  } while (0)
  if (N == 19) GG_R5(19);
  else if (N == 13) GG_R5(13);
  else GG_R5(9);
#undef GG_R5
}

}  // namespace gg

#ifdef GG_AB_SWEEPS
// A/B builds only: read and clear the sweep counters of k_rollout5's flood batches
extern "C" int32_t gg_ab_sweeps_read_r5(unsigned long long *out2) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out2, HIP_SYMBOL(gg::gg_sweeps), 16) != hipSuccess) return 2;
  unsigned long long z[2] = {0, 0};
  return hipMemcpyToSymbol(HIP_SYMBOL(gg::gg_sweeps), z, 16) == hipSuccess ? 0 : 3;
}
#endif

#ifdef GG_AB_PROF
// A/B builds only: read and clear the phase clocks of THIS translation unit's launches (gg_prof has internal linkage)
GG_PROF_READ(gg_ab_prof_read_r5)
#endif
