// gg_rollout.hip - the launches of the fused multi-ply kernel with drawn moves (gg_batch_rollout, gg_batch_rollout_packed,
// gg_batch_rollout_tracked: k_rollout4<R, IO, false, FULLN>) as a translation unit of their own, because it is compiled with
// ONE different code-generation switch: -mllvm -enable-post-misched=false (Makefile).  The ply of this kernel is a long
// straight-line dependency chain that the source already orders; the post-register-allocation scheduler's reordering costs
// it 1.0 % (1.881 -> 1.861 ms per 256-ply launch of 65 536 games, identical states), while the same switch on the rest of
// the library loses 1 - 4 % on the children, env-step and replay kernels (round 4, A/B builds on one box) - so only these
// instantiations get it.  Everything else about the kernel lives in gg_v4.h; argument checks, device selection and grid
// sizing stay in gg_kernels.hip, which calls launch_rollout4().
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gg_common.h"
#include "gg_v2.h"
#include "gg_v4.h"
#include "gg_ws.h"   // (defines the weighted-draw helpers k_rollout4 names in its WTS branch, dead here)

namespace gg {

// the instantiation with compile-time N when the board fills its row capacity (9, 13, 19)
#define GG_ROLLOUT4(IO)                                                                                                     \
  do {                                                                                                                      \
    if (N == 9) k_rollout4<9, IO, false, true><<<grid, kWave, 0, s>>>(st, rng, last_actions, steps_done, B, N, inv, plies, auto_reset, nb);        \
    else if (N < 9) k_rollout4<9, IO, false, false><<<grid, kWave, 0, s>>>(st, rng, last_actions, steps_done, B, N, inv, plies, auto_reset, nb);   \
    else if (N == 13) k_rollout4<13, IO, false, true><<<grid, kWave, 0, s>>>(st, rng, last_actions, steps_done, B, N, inv, plies, auto_reset, nb); \
    else if (N < 13) k_rollout4<13, IO, false, false><<<grid, kWave, 0, s>>>(st, rng, last_actions, steps_done, B, N, inv, plies, auto_reset, nb); \
    else if (N == 19) k_rollout4<19, IO, false, true><<<grid, kWave, 0, s>>>(st, rng, last_actions, steps_done, B, N, inv, plies, auto_reset, nb); \
    else k_rollout4<19, IO, false, false><<<grid, kWave, 0, s>>>(st, rng, last_actions, steps_done, B, N, inv, plies, auto_reset, nb);             \
  } while (0)

// io: 0 byte planes, 1 packed boards, 2 tracked boards (`st` is the batch in that format)
void launch_rollout4(int io, uint8_t *st, uint64_t *rng, int32_t *last_actions, int64_t *steps_done, int64_t B, int32_t N,
                     uint32_t inv, int plies, int auto_reset, int nb, int grid, hipStream_t s) {
  if (io == 0) GG_ROLLOUT4(0);
  else if (io == 1) GG_ROLLOUT4(1);
  else GG_ROLLOUT4(2);
}
#undef GG_ROLLOUT4

}  // namespace gg

#ifdef GG_AB_PROF
// A/B builds only: read and clear the phase clocks of THIS translation unit's k_rollout4 launches (gg_prof has internal linkage)
GG_PROF_READ(gg_ab_prof_read_rollout)
#endif
