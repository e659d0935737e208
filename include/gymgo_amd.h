/*
 * gymgo_amd.h - C-ABI of the MI355X (gfx950) batched Go step path.
 *
 * The reference (huangeddie/GymGo) is pure Python and has no FFI seam; its boundary for this path
 * is the function API of gym_go/gogame.py + gym_go/state_utils.py.  Each entry point below is what
 * a ctypes binding of that API binds to (the stub is shown in INTEGRATION.md); the reference
 * function it replaces is cited as path:line relative to the reference root.
 *
 * Conventions (all entry points):
 *   - States are contiguous uint8 [B][6][N][N], values in {0,1}; channels per gym_go/govars.py:4-9
 *     (0 black, 1 white, 2 turn, 3 invalid moves for the side to move, 4 previous move was a pass,
 *     5 game over).  Planes 2, 4 and 5 are uniform by construction (the reference only ever writes
 *     them whole: gym_go/gogame.py:49-56, gym_go/state_utils.py:241); the kernels read one byte of each.
 *   - Every pointer is a DEVICE pointer owned by the caller; the library allocates nothing, reads one environment
 *     variable (GYMGO_AMD_CUS, see gg_device_cus: performance only) and never synchronises: work is enqueued on `hip_stream` (a hipStream_t of the device that owns the
 *     buffers, NULL = its default stream) and the call returns immediately.  The kernels run on the device that owns
 *     the first buffer argument, whatever the calling thread's current device is (it is restored before the call
 *     returns).
 *   - Global state.  No result depends on anything but the arguments.  The library keeps three pieces of mutable state,
 *     all performance-only: a per-device cache of the CU count (relaxed atomics; racing first callers store the same value), a
 *     per-(device, kernel) cache of kernel occupancy (behind a mutex) and, in device memory, the "FairShare" progress
 *     boards of the fused multi-ply kernels (two of 512 KB per device - one per translation unit of the library: the
 *     fused rollouts with drawn moves, and the replay / env-step launches; waves of the other unit's kernels are
 *     "foreign" to a board, like another stream's - every wave of such a launch publishes the ply it has reached and
 *     reads its SIMD-mates' words to set its own issue priority; stale or foreign words only shift priorities).  Every entry point is re-entrant and
 *     thread-safe: concurrent calls from several threads on several streams are supported
 *     (tests/test_gpu_threads.py; tools/sanitize.sh: the host side under ASan / UBSan / TSan).
 *   - CPU twins.  SURVEY 8(b) sketched `_cpu`-suffixed entry points with host pointers next to these.  They are
 *     deliberately NOT exported: the CPU restatement of the path is test infrastructure (oracle/gg_oracle.c,
 *     `gg_oracle_*`, linked by tests and the bench's cpu_baseline only), and a product library that could fall back
 *     to it would void every parity claim.  A missing GPU is an error (hipErrorNoDevice / GymGoNativeError).
 *   - Which kernel serves a call depends on its arguments only (board size, batch size, plies per launch).
 *   - Alignment.  Boards may start at any byte offset, but HBM is only touched with naturally aligned 16-byte accesses:
 *     a state / children / tracked buffer must be READABLE from its start rounded down to 16 bytes to its end rounded
 *     up to 16 bytes (always true for a whole allocation and for any slice of one; bytes outside the buffer are read,
 *     never written).
 *   - 2 <= N <= 19.  Actions are int32 in [0, N*N]; N*N = pass (gym_go/gogame.py:40-42).
 *   - Return value: 0 on success, a hipError_t (> 0) for launch/runtime errors, or a negative
 *     GG_E_* code for bad arguments.  Re-entrant; safe from several threads / one process per GPU.
 */
#ifndef GYMGO_AMD_H
#define GYMGO_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GG_ABI_VERSION 5
#define GG_MAX_BOARD 19
#define GG_NUM_CHNLS 6

#define GG_E_BADSIZE (-1)  /* N outside [2, 19] or B < 0 */
#define GG_E_NULLPTR (-2)  /* a required pointer is NULL */
#define GG_E_BADARG (-3)   /* other argument out of range */

/* element type of policy weights (gg_batch_sample_weighted*, gg_batch_env_step_tracked_weighted) */
#define GG_W_F32 0  /* float32 */
#define GG_W_BF16 1 /* bfloat16: the upper half of a float32 */
#define GG_W_F16 2  /* IEEE float16 */

/* per-game status written by gg_batch_next_states */
#define GG_STATUS_OK 0
#define GG_STATUS_ILLEGAL 1 /* point has INVD set / action out of range: the reference raises
                               AssertionError (gym_go/gogame.py:59, :117); the row is copied through */

int32_t gg_version(void);

/* Number of compute units the library sizes its grids for: those of the calling thread's current device (0 if no device),
   or the value of the environment variable GYMGO_AMD_CUS (1 ... 4096, read once per process) - for a partition of the GPU
   or a GPU shared with other work.  Results never depend on it (tests/test_gpu_cus.py). */
int32_t gg_device_cus(void);

/*
 * gogame.batch_next_states(batch_states, batch_action1d, canonical)   gym_go/gogame.py:90-150
 * with the per-game semantics of gogame.next_state                      gym_go/gogame.py:34-87
 * (place / pass, state_utils.update_pieces capture resolution :159-180, ko :72-75,
 * state_utils.compute_invalid_moves :24-83, state_utils.set_turn :235-241, optional
 * canonical_form :313-321) for EVERY game - the reference's batch path mis-aligns games when the
 * batch contains passes (gym_go/state_utils.py:187-193); that defect is not reproduced.
 * `in` and `out` must not overlap.  `status` may be NULL.
 */
int32_t gg_batch_next_states(const uint8_t *in, const int32_t *actions, uint8_t *out, int32_t *status,
                             int64_t B, int32_t N, int32_t canonical, void *hip_stream);

/*
 * gogame.batch_next_states with a caller-owned WORKSPACE                gym_go/gogame.py:90-150
 * Same arguments, results and status as gg_batch_next_states, plus `workspace`: uint32 [B][gg_tracked_words(N)], zero-filled
 * before its first use and otherwise opaque.  The call leaves in it the tracked form (stones + liberty classes) of every
 * position it wrote to `out`; the next call takes the liberty classes of game b from there when planes 0 / 1 of in[b]
 * equal the workspace's stones EXACTLY (checked per board, every call) and analyses the board from scratch otherwise.
 * A loop that feeds each output batch back as the next input - a rollout through the step API - therefore pays the
 * full liberty analysis once; results never depend on the workspace content.
 */
int32_t gg_batch_next_states_ws(const uint8_t *in, const int32_t *actions, uint8_t *out, int32_t *status, uint32_t *workspace,
                                int64_t B, int32_t N, int32_t canonical, void *hip_stream);

/*
 * state_utils.batch_compute_invalid_moves                              gym_go/state_utils.py:86-156
 * Recomputes plane 3 (invalid moves for the side to move, plane 2) from planes 0-2:
 * mask[b] = compute_invalid_moves(states[b], player = 1 - turn(states[b]), ko[b]).
 * `ko` (nullable) holds a flat point index per game or -1.  mask is uint8 [B][N][N].
 */
int32_t gg_batch_invalid_mask(const uint8_t *states, const int32_t *ko, uint8_t *mask, int64_t B, int32_t N,
                              void *hip_stream);

/*
 * gogame.batch_areas                                                   gym_go/gogame.py:303-310
 * (gogame.areas :275-300, Tromp-Taylor): black[b], white[b] = area of each colour, as int32
 * (the reference returns the same integers as float64).
 */
int32_t gg_batch_areas(const uint8_t *states, int32_t *black, int32_t *white, int64_t B, int32_t N,
                       void *hip_stream);

/*
 * gogame.children(state, canonical, padded=True) for every state      gym_go/gogame.py:175-186
 * children is uint8 [B][N*N+1][6][N][N]; slot a = next_state(states[b], a, canonical) when action a
 * is valid (plane 3 clear, or a = pass), all zeros otherwise (gym_go/tests/test_basics.py:209-223).
 */
int32_t gg_batch_children(const uint8_t *states, uint8_t *children, int64_t B, int32_t N, int32_t canonical,
                          void *hip_stream);

/* gogame.children(state, canonical, padded=False) (gym_go/gogame.py:175-180: the un-padded result is what the reference
 * computes first, :179; GoEnv.children forwards the flag, gym_go/envs/go_env.py:105-109) for a whole batch, in two calls:
 *   1. gg_batch_children_offsets: offsets int32 [B+1] = exclusive prefix sums of the number of children valid_moves() keeps
 *      per parent (:153-161: the points whose plane-3 byte is 0, + the pass; ALL N*N+1 actions once the game has ended),
 *      offsets[B] = the total.  The caller reads offsets[B] back and allocates children: uint8 [offsets[B]][6][N][N].
 *      order (int32 [B], may be NULL): the parents sorted by falling child count - the order in which the expansion should
 *      hand them to the machine (a parent's work grows with its children; heaviest first keeps the launch's tail short).
 *      GG_E_BADSIZE if B * (N*N+1) does not fit an int32.
 *   2. gg_batch_children_compact: parent b's children at children[offsets[b] .. offsets[b+1]), ascending action order - exactly
 *      the non-zero-padded slots of gg_batch_children, i.e. children_padded[b][valid_moves(states[b]) == 1] (a kept slot whose
 *      move the reference would have refused - an ended game, plane 3 clear on a stone - is all zero, as in the padded form).
 *      order: what gg_batch_children_offsets wrote for the same states (any permutation of 0 .. B-1 gives the same result),
 *      or NULL: index order.
 * On mid-game 19x19 parents a third of the 362 slots is kept: a third of the bytes of the padded expansion. */
int32_t gg_batch_children_offsets(const uint8_t *states, int32_t *offsets, int32_t *order, int64_t B, int32_t N, void *hip_stream);
int32_t gg_batch_children_compact(const uint8_t *states, const int32_t *offsets, const int32_t *order, uint8_t *children, int64_t B,
                                  int32_t N, int32_t canonical, void *hip_stream);

/*
 * Uniform-random rollout, `plies` steps per game, IN PLACE, board resident on-chip between plies:
 * per ply and game  a ~ Uniform{valid actions incl. pass}  (GoEnv.uniform_random_action,
 * gym_go/envs/go_env.py:78-81; gogame.random_action gym_go/gogame.py:395-404), then
 * state = next_state(state, a).  A finished game (plane 5 set) is reset to zeros first when
 * auto_reset != 0, otherwise it is left frozen and draws nothing.
 * rng: uint64 [B] per-game generator state (see gg_rng_seed), advanced once per ply played.
 * last_actions (nullable): int32 [B], the last action applied during this call (-1 if the game was frozen throughout).
 * steps_done (nullable): int64 [B], incremented by the number of plies actually played.
 * Sampler (build-defined, mirrored by oracle/gg_oracle.c): x += 0x9E3779B97F4A7C15;
 * u = splitmix64_finalise(x); k = ((u >> 32) * n_valid) >> 32; action = k-th valid action ascending.
 */
int32_t gg_batch_rollout(uint8_t *states, uint64_t *rng, int32_t *last_actions, int64_t *steps_done, int64_t B,
                         int32_t N, int32_t plies, int32_t auto_reset, void *hip_stream);

/*
 * GoEnv.step for every game of a batched env, IN PLACE, one launch            gym_go/envs/go_env.py:49-76
 *   1. auto_reset != 0: a finished game (plane 5 set) is reset first (GoEnv.reset, :40-47); auto_reset == 0: it
 *      is refused (status 1, row untouched; the reference asserts `not self.done`, :53).
 *   2. the action: actions[b] (0 .. N*N, N*N = pass) or, when actions == NULL, drawn uniformly over the valid
 *      actions with rng[b] exactly like gg_batch_rollout / gg_batch_sample_actions (uniform_random_action, :78-81).
 *   3. legality (gogame.py:59): out of range or on a set point of plane 3 -> status 1, row untouched.
 *   4. state = gogame.next_state(state, action)  (:34-87), dones[b] = game_ended (:189-196).
 *   5. rewards[b] = GoEnv.reward() (:128-149) from black's perspective with Tromp-Taylor areas (gogame.py:275-300)
 *      of the resulting position: GG_REWARD_REAL  done ? sign(black - white - komi) : 0;
 *      GG_REWARD_HEURISTIC  done ? (margin > 0 ? +N*N : -N*N) : margin.
 * rng: uint64 [B], required when actions == NULL (advanced once per game that draws).  rewards float32 [B], dones
 * uint8 [B], status int32 [B], taken_actions int32 [B] (the action used): each nullable.
 */
#define GG_REWARD_REAL 0
#define GG_REWARD_HEURISTIC 1
int32_t gg_batch_env_step(uint8_t *states, const int32_t *actions, uint64_t *rng, float *rewards, uint8_t *dones,
                          int32_t *status, int32_t *taken_actions, int64_t B, int32_t N, float komi,
                          int32_t reward_method, int32_t auto_reset, void *hip_stream);

/*
 * gg_batch_env_step + the score of every resulting position, ONE launch         gym_go/envs/go_env.py:49-76, :128-149
 * areas: int32 [B][2] (required) = gogame.areas (gym_go/gogame.py:275-300) of the position each game is left in: black, white -
 * what GoEnv.reward / winning / winner read after the step.  Everything else as gg_batch_env_step (the reward follows
 * `reward_method`).  GoEnv.step of the Python package is this call at B = 1 on a record in pinned, device-mapped host
 * memory (hipHostMalloc: the kernel reads the action and writes state, areas, status and done in place - no copy either way);
 * like every entry point it accepts any device-accessible pointer whose 16-byte-aligned superset is readable.
 */
int32_t gg_batch_env_step_scored(uint8_t *states, const int32_t *actions, uint64_t *rng, float *rewards, uint8_t *dones,
                                 int32_t *status, int32_t *taken_actions, int32_t *areas, int64_t B, int32_t N, float komi,
                                 int32_t reward_method, int32_t auto_reset, void *hip_stream);

/*
 * One sampling pass only (no step): actions[b] ~ Uniform{valid actions of states[b] incl. pass},
 * same generator as gg_batch_rollout (advances rng[b] once).  Finished games: reset is NOT applied;
 * every action counts as valid there (gogame.invalid_moves returns zeros once ended, :155-156).
 */
int32_t gg_batch_sample_actions(const uint8_t *states, uint64_t *rng, int32_t *actions, int64_t B, int32_t N,
                                void *hip_stream);

/*
 * state_utils.update_pieces / batch_update_pieces                       gym_go/state_utils.py:159-211
 * Stand-alone capture resolution (inside gg_batch_next_states it is fused), with the reference's own inputs:
 * adj is int32 [B][K] - the locations whose opponent groups are examined, as flat indices r * N + c (the reference
 * passes the on-board neighbours of the stone just placed: adj_locs of state_utils.adj_data, :214-223, so K = 4;
 * entries outside [0, N*N) are unused) - and players[b] is the side that moved.  Every group of the OTHER colour that
 * holds one of these locations and has no empty point next to it - liberties are taken on the position as given,
 * before any removal, like `empties` at :164 - is removed IN PLACE (planes 0/1) and marked in killed
 * (uint8 [B][N][N], nullable).  The position need not be reachable by legal play.
 */
int32_t gg_batch_update_pieces(uint8_t *states, const int32_t *adj, int32_t K, const int32_t *players, uint8_t *killed,
                               int64_t B, int32_t N, void *hip_stream);

/*
 * Auto-reset of a batched env (build-side policy; the reference has one game per GoEnv and resets by hand,
 * gym_go/envs/go_env.py:40-47): every game whose game-over plane is set becomes gogame.init_state (all zeros).
 */
int32_t gg_batch_reset_finished(uint8_t *states, int64_t B, int32_t N, void *hip_stream);

/*
 * Bit-packed state format for replay buffers / checkpoints / the wire (no reference counterpart; the reference
 * stores 0/1 in float64, gym_go/gogame.py:22-25).  One board = gg_packed_words(N) = 3 N + 1 uint32:
 * N row masks (bit c = column c) of plane 0, of plane 1, of plane 3, then a flag word (bit 0 turn, bit 1 previous
 * move was a pass, bit 2 game over).  unpack(pack(s)) == s for every state whose planes 2/4/5 are uniform.
 */
int32_t gg_packed_words(int32_t N);
int32_t gg_batch_pack_states(const uint8_t *states, uint32_t *packed, int64_t B, int32_t N, void *hip_stream);
int32_t gg_batch_unpack_states(const uint32_t *packed, uint8_t *states, int64_t B, int32_t N, void *hip_stream);

/*
 * The step path on PACKED boards (uint32 [B][gg_packed_words(N)], the format of gg_batch_pack_states): same semantics,
 * arguments and error behaviour as the byte-plane entry points of the same name, with `packed` in place of `states`.
 * A board is 232 B instead of 2 166 B at 19x19 and the kernels skip the byte <-> bit conversions - for search trees and
 * replay buffers that keep states packed and unpack only what goes to a network.
 *   gg_batch_next_states_packed   gogame.batch_next_states         gym_go/gogame.py:90-150   (in / out must not overlap)
 *   gg_batch_rollout_packed       loop of uniform_random_action + step  gym_go/envs/go_env.py:49-81   (in place)
 *   gg_batch_env_step_packed      GoEnv.step + reward + reset      gym_go/envs/go_env.py:40-76, :128-149   (in place)
 *   gg_batch_children_packed      gogame.children per state        gym_go/gogame.py:175-186
 *                                 children: uint32 [B][N*N+1][gg_packed_words(N)], slots of invalid actions all zero
 */
int32_t gg_batch_next_states_packed(const uint32_t *in, const int32_t *actions, uint32_t *out, int32_t *status, int64_t B,
                                    int32_t N, int32_t canonical, void *hip_stream);
int32_t gg_batch_rollout_packed(uint32_t *packed, uint64_t *rng, int32_t *last_actions, int64_t *steps_done, int64_t B,
                                int32_t N, int32_t plies, int32_t auto_reset, void *hip_stream);
int32_t gg_batch_env_step_packed(uint32_t *packed, const int32_t *actions, uint64_t *rng, float *rewards, uint8_t *dones,
                                 int32_t *status, int32_t *taken_actions, int64_t B, int32_t N, float komi,
                                 int32_t reward_method, int32_t auto_reset, void *hip_stream);
int32_t gg_batch_children_packed(const uint32_t *packed, uint32_t *children, int64_t B, int32_t N, int32_t canonical,
                                 void *hip_stream);

/*
 * Replay of given move sequences, IN PLACE, the boards resident on-chip for all T moves (one launch instead of T):
 * for t = 0 .. T-1: states[b] = gogame.next_state(states[b], moves[b][t])      gym_go/gogame.py:34-87
 * i.e. a loop of GoEnv.step (gym_go/envs/go_env.py:49-76) over recorded games, search lines or a policy's action buffer.
 * moves: int32 [B][T] (N*N = pass).  A game stops at its first move that is out of range, on an invalid point
 * (gogame.py:59) or made after the game has ended (go_env.py:53) and keeps the state before that move;
 * played (nullable): int32 [B] = number of moves applied (T if all were).  _packed: the same on packed boards.
 */
int32_t gg_batch_play_moves(uint8_t *states, const int32_t *moves, int32_t *played, int64_t B, int32_t N, int32_t T,
                            void *hip_stream);
int32_t gg_batch_play_moves_packed(uint32_t *packed, const int32_t *moves, int32_t *played, int64_t B, int32_t N, int32_t T,
                                   void *hip_stream);

/*
 * TRACKED boards: uint32 [B][gg_tracked_words(N) = 5 N + 1] = the packed board (rows of planes 0 / 1 / 3) + two more
 * row sets - the black / the white stones whose group has >= 2 liberties - + the flag word (bit 0 turn, bit 1 previous
 * move was a pass, bit 2 game over).  A board that carries its liberty classes needs no analysis when a launch starts,
 * so stepping it ONE ply per launch (a policy network choosing every move) runs at the fused kernel's rate.
 *   gg_batch_track_states       uint8 [B][6][N][N] -> tracked (classes by one analysis)
 *   gg_batch_untrack_states     tracked -> uint8 [B][6][N][N]
 *   gg_batch_rollout_tracked    as gg_batch_rollout, in place on tracked boards
 *   gg_batch_play_moves_tracked as gg_batch_play_moves, in place on tracked boards (T = 1: one GoEnv.step per game)
 * The class rows must belong to the position: boards edited by the caller go through untrack / track again.
 */
int32_t gg_tracked_words(int32_t N);
int32_t gg_batch_track_states(const uint8_t *states, uint32_t *tracked, int64_t B, int32_t N, void *hip_stream);
int32_t gg_batch_untrack_states(const uint32_t *tracked, uint8_t *states, int64_t B, int32_t N, void *hip_stream);
int32_t gg_batch_rollout_tracked(uint32_t *tracked, uint64_t *rng, int32_t *last_actions, int64_t *steps_done, int64_t B,
                                 int32_t N, int32_t plies, int32_t auto_reset, void *hip_stream);
int32_t gg_batch_play_moves_tracked(uint32_t *tracked, const int32_t *moves, int32_t *played, int64_t B, int32_t N, int32_t T,
                                    void *hip_stream);

/*
 * GoEnv.step for every game of a batched env whose boards are kept TRACKED, IN PLACE, one launch
 * (gym_go/envs/go_env.py:49-76): the same steps 1-5, arguments and outputs as gg_batch_env_step, on boards that carry
 * their liberty classes - no per-ply analysis, so a policy-driven env steps at the multi-ply kernel's per-ply rate.
 * states_out (nullable): uint8 [B][6][N][N], the resulting position of EVERY game as byte planes - the observation
 * GoEnv.step returns (:66), written by the same launch (pass NULL to keep only the tracked boards current).
 * steps_done (nullable): int64 [B], += 1 for every game whose step was played (status GG_STATUS_OK).
 * GG_REWARD_REAL scores a game only when it ends (a rare path); GG_REWARD_HEURISTIC scores every game every step.
 */
int32_t gg_batch_env_step_tracked(uint32_t *tracked, const int32_t *actions, uint64_t *rng, float *rewards, uint8_t *dones,
                                  int32_t *status, int32_t *taken_actions, uint8_t *states_out, int64_t *steps_done,
                                  int64_t B, int32_t N, float komi, int32_t reward_method, int32_t auto_reset,
                                  void *hip_stream);

/*
 * gg_batch_env_step_tracked with the move of every game DRAWN FROM POLICY WEIGHTS by the same launch:
 * gogame.random_weighted_action (gym_go/gogame.py:385-392) fused into GoEnv.step (gym_go/envs/go_env.py:49-76).
 * weights: [B][N*N+1] of float32 / bfloat16 / float16 (weight_dtype = GG_W_*; one weight per action, the pass last; the
 * 16-bit forms are widened exactly and halve the bytes this launch reads).  Per game: a finished game is reset first when
 * auto_reset; the weights are masked by the game's invalid-move rows (the pass is always playable), L1-normalised and
 * drawn from as gg_batch_sample_weighted describes, with rng[b] (which advances once; a frozen game - finished,
 * auto_reset == 0 - draws nothing and is refused).  A game whose playable weights are all zero is refused
 * (status GG_STATUS_ILLEGAL, taken action -1) - np.random.choice raises for such a vector.  Other arguments and
 * outputs as gg_batch_env_step_tracked; taken_actions receives the drawn moves.
 */
int32_t gg_batch_env_step_tracked_weighted(uint32_t *tracked, const void *weights, int32_t weight_dtype, uint64_t *rng, float *rewards,
                                           uint8_t *dones, int32_t *status, int32_t *taken_actions, uint8_t *states_out,
                                           int64_t *steps_done, int64_t B, int32_t N, float komi, int32_t reward_method,
                                           int32_t auto_reset, void *hip_stream);

/*
 * gogame.random_weighted_action(move_weights)                           gym_go/gogame.py:385-392
 * gogame.random_action(state) = the same with weights 1 - invalid       gym_go/gogame.py:395-404
 * for every game: actions[b] ~ weights[b] / sum(weights[b]) over the playable actions.  The reference normalises in
 * float64 and draws from NumPy's global generator; so that device and oracle agree bit for bit the draw is defined in
 * integers: (1) each weight (float32, or bfloat16 / float16 widened exactly: weight_dtype = GG_W_*) is clamped to
 * [+0, FLT_MAX] on its float32 bit pattern (anything with the sign bit set, -NaN included -> 0; +NaN / +inf -> FLT_MAX)
 * and zeroed where plane 3 of states[b] is set (the reference ASSUMES invalid moves have weight 0, :387; the pass is
 * never masked, a finished game masks nothing, gym_go/gogame.py:155-156; states == NULL: no mask); (2) with E = max(the
 * largest weight's biased exponent, 24), q[a] = trunc(w[a] * 2^(148 - E)) < 2^22: the weights as 22-bit fixed point
 * relative to the largest; (3) T = sum q, k = floor((u >> 32) * T / 2^32) with u the next output of rng[b]
 * (gg_rng_seed's generator, advanced once per game per call); (4) the action is the first one, in the interleaved order
 * a = i + 16 j (i = 0..15 outer), whose running sum of q exceeds k, so P(a) = q[a] / T.  T == 0 gives actions[b] = -1.
 */
int32_t gg_batch_sample_weighted(const uint8_t *states, const void *weights, int32_t weight_dtype, uint64_t *rng,
                                 int32_t *actions, int64_t B, int32_t N, void *hip_stream);

/* The same draw for row-mask boards: planes = 3 (packed, gg_batch_pack_states) or 5 (tracked, gg_batch_track_states). */
int32_t gg_batch_sample_weighted_rows(const uint32_t *boards, int32_t planes, const void *weights, int32_t weight_dtype,
                                      uint64_t *rng, int32_t *actions, int64_t B, int32_t N, void *hip_stream);

/*
 * gogame.all_symmetries(image) / gogame.random_symmetry(image)          gym_go/gogame.py:340-382
 * for a batch: in is uint8 [B][C][N][N] (any C >= 1 with C*N*N <= 8192: states, observations, per-point targets).
 * orient: int32 [B], the orientation of each game in 0..7 composed exactly as the reference does (bit 0 flip the columns,
 * then bit 1 flip the rows, then bit 2 np.rot90 over the board axes) -> out is [B][C][N][N]; orient == NULL: all eight
 * views of every game, out is [B][8][C][N][N] in the order of all_symmetries.  in and out must not overlap.
 */
int32_t gg_batch_symmetry(const uint8_t *in, const int32_t *orient, uint8_t *out, int64_t B, int32_t C, int32_t N,
                          void *hip_stream);

/*
 * The same on row-mask boards (planes = 3 packed / 5 tracked; uint32 [B][planes*N+1]): every row plane is transformed
 * (a column flip is a bit reversal, a row flip a row permutation, the rotation a bit transpose), the flag word copied -
 * liberty classes and the invalid-move rows (ko point included) are geometric, so the result is a valid board of the
 * same format.  out is [B][W] (orient given) or [B][8][W].
 */
int32_t gg_batch_symmetry_rows(const uint32_t *in, int32_t planes, const int32_t *orient, uint32_t *out, int64_t B,
                               int32_t N, void *hip_stream);

/* rng[b] = initial generator state for (base_seed, game index first_game + b). */
int32_t gg_rng_seed(uint64_t *rng, uint64_t base_seed, int64_t first_game, int64_t B, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* GYMGO_AMD_H */
