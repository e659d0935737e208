#!/usr/bin/env python
"""Per-entry-point throughput of the C-ABI on one GPU (BASELINE.json configs 2, 3, 5 + the small ops), each
against its own HBM roofline (algorithmic bytes per unit).  Prints one JSON object; run on the GPU box:
    python tools/bench_ops.py > gpurun_out/ops.json
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gymgo_amd import _lib  # noqa: E402
if os.environ.get('LIB'):      # A/B: another build of the library (path relative to the repo root), e.g. LIB=ab_libs/libgg_x.so
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ['LIB'])
from gymgo_amd import gogame, state_utils  # noqa: E402

PEAK = 8.0e12


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def midgame(B, N, plies, seed=3):
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, seed)
    gogame.batch_rollout(st, rng, plies, True)
    return st, rng


def main():
    out = {}
    for N, B, plies in ((9, 4096, 60), (19, 65536, 250)):
        st, rng = midgame(B, N, plies)
        S = 6 * N * N
        # out-of-place step API fed by the on-device sampler (finished games reset by a masked fill, as a user would)
        acts = gogame.batch_sample_actions(st, rng)

        def api_ply():
            nonlocal st
            gogame.batch_reset_finished(st)
            a = gogame.batch_sample_actions(st, rng)
            st, _ = gogame.batch_next_states(st, a, check=False)
        t = timed(api_ply, 50)
        out['vecenv_step_api_%dx%d_B%d' % (N, N, B)] = {'steps_per_s': B / t, 'note': 'reset_finished + sample_actions + next_states per ply'}
        from gymgo_amd.envs import GoVecEnv
        for method in ('real', 'heuristic'):
            for layout in ('tracked', 'bytes'):
                env = GoVecEnv(B, N, komi=7.5, reward_method=method, layout=layout)
                env.rollout(plies)
                t = timed(lambda: env.step(), 50)
                # the bytes THIS launch moves per game: a tracked board in and out (2 x 4 (5 N + 1) B) + the observation (S B) + the
                # generator and the per-game outputs (25 B); in place on byte planes: the board in and out + the same 25 B.
                # (round 5 priced the tracked step against the 2 S + 29 B of an out-of-place byte-plane step: 0.915 instead of 0.62)
                moved = (8 * (5 * N + 1) + S + 25) if layout == 'tracked' else (2 * S + 25)
                out['GoVecEnv_step_%s_reward_%s_%dx%d_B%d' % (method, layout, N, N, B)] = {
                    'steps_per_s': B / t, 'bytes_moved_per_step': moved, 'moved_GBps': B * moved / t / 1e9, 'hbm_frac': B * moved / t / PEAK,
                    'note': 'one launch: sample + auto-reset + step + areas + rewards + dones + the uint8 observation (tracked: '
                            'gg_batch_env_step_tracked on resident tracked boards; bytes: gg_batch_env_step in place)'}
            env = GoVecEnv(B, N, komi=7.5, reward_method=method, layout='bytes')
            env.rollout(plies)
            t = timed(lambda: env.step_unfused(env.sample_actions()), 30)
            out['GoVecEnv_step_unfused_%s_reward_%dx%d_B%d' % (method, N, N, B)] = {'steps_per_s': B / t, 'note': 'same step as separate launches'}
        st2, _ = midgame(B, N, plies, 5)
        acts = gogame.batch_sample_actions(st2, gogame.rng_seed(B, 9))
        t = timed(lambda: gogame.batch_next_states(st2, acts, check=False), 50)
        out['gg_batch_next_states_%dx%d_B%d' % (N, N, B)] = {
            'steps_per_s': B / t, 'algorithmic_GBps': B * (2 * S + 4) / t / 1e9, 'roofline_frac': B * (2 * S + 4) / t / PEAK}
        t = timed(lambda: gogame.batch_next_states(st2, acts, canonical=True, check=False), 50)
        out['gg_batch_next_states_canonical_%dx%d_B%d' % (N, N, B)] = {'steps_per_s': B / t}
        t = timed(lambda: gogame.batch_areas(st2), 50)
        out['gg_batch_areas_%dx%d_B%d' % (N, N, B)] = {
            'boards_per_s': B / t, 'bytes_per_board': 2 * N * N + 8, 'moved_GBps': B * (2 * N * N + 8) / t / 1e9,
            'roofline_frac': B * (2 * N * N + 8) / t / PEAK,
            'note': 'the kernel reads planes 0 / 1 only (2 N^2 of the 6 N^2 bytes of a state) and writes two int32 per board'}
        t = timed(lambda: state_utils.batch_compute_invalid_moves(st2, None, None), 50)
        out['gg_batch_invalid_mask_%dx%d_B%d' % (N, N, B)] = {'boards_per_s': B / t}
        r2 = gogame.rng_seed(B, 11)
        t = timed(lambda: gogame.batch_sample_actions(st2, r2), 50)
        out['gg_batch_sample_actions_%dx%d_B%d' % (N, N, B)] = {'boards_per_s': B / t}
    # config 5: 19x19, 8192 parents, full 362-slot padded expansion
    N, B = 19, 8192
    st, _ = midgame(B, N, 250, 7)
    S = 6 * N * N
    # config 5 through THE harness of the bench line (bench.children_record: one preallocated buffer, HIP events) - padded and
    # un-padded - on this script's parents (250 plies, one phase) next to the bench line's stationary mix
    import bench
    out['gg_batch_children_19x19_B8192'] = bench.children_record(torch, torch.device('cuda', 0), st, 'midgame(8192, 19, 250 plies, seed 7)', by_phase=True)
    out['gg_batch_children_19x19_B8192']['mean_valid_children'] = float(gogame.batch_valid_moves(st).float().sum() / B)
    t = timed(lambda: gogame.batch_children(st, canonical=True), 5)
    out['gg_batch_children_19x19_B8192_canonical_with_allocation_per_call'] = {'parents_per_s': B / t, 'ms_per_batch': t * 1e3}
    # the same parents as packed boards: 362 x 232 B per parent
    pk = gogame.batch_pack(st)
    t = timed(lambda: gogame.batch_children_packed(pk), 10)
    out['gg_batch_children_packed_19x19_B8192'] = {
        'parents_per_s': B / t, 'child_states_per_s': B * (N * N + 1) / t, 'ms_per_batch': t * 1e3,
        'bytes_per_parent': (N * N + 2) * (3 * N + 1) * 4, 'moved_GBps': B * (N * N + 2) * (3 * N + 1) * 4 / t / 1e9}
    # packed step path, 65 536 games
    B2 = 65536
    st2, rng2 = midgame(B2, N, 250, 5)
    pk2 = gogame.batch_pack(st2)
    acts = gogame.batch_sample_actions(st2, rng2)
    PB = (3 * N + 1) * 4
    t = timed(lambda: gogame.batch_next_states_packed(pk2, acts, check=False), 50)
    out['gg_batch_next_states_packed_19x19_B65536'] = {'steps_per_s': B2 / t, 'moved_GBps': B2 * (2 * PB + 4) / t / 1e9}
    buf = None

    def pstep(method):
        nonlocal buf
        buf = gogame.batch_env_step_packed(pk2, None, rng2, 7.5, method, True, out=buf)
    for method in ('real', 'heuristic'):
        t = timed(lambda: pstep(method), 50)
        out['gg_batch_env_step_packed_%s_19x19_B65536' % method] = {'steps_per_s': B2 / t}
    for F in (1, 64):
        t = timed(lambda: gogame.batch_rollout_packed(pk2, rng2, F, True), 20 if F == 1 else 5)
        out['gg_batch_rollout_packed_F%d_19x19_B65536' % F] = {'steps_per_s': B2 * F / t}
    # tracked boards (packed + liberty classes): one ply per launch at the fused kernel's rate
    tr = gogame.batch_track(st2)
    rt = gogame.rng_seed(B2, 31)
    played = torch.empty(B2, dtype=torch.int32, device='cuda')
    for F in (1, 64):
        t = timed(lambda: gogame.batch_rollout_tracked(tr, rt, F, True), 50 if F == 1 else 5)
        out['gg_batch_rollout_tracked_F%d_19x19_B65536' % F] = {'steps_per_s': B2 * F / t}

    def policy_step():   # the policy-driven step: on-device sampler standing in for a network, then one move per game
        a = gogame.batch_sample_actions(gogame.batch_untrack(tr), rt)
        gogame.batch_play_moves_tracked(tr, a, played)
    one = gogame.batch_sample_actions(gogame.batch_untrack(tr), rt)
    t = timed(lambda: gogame.batch_play_moves_tracked(tr, torch.full_like(one, N * N), played), 50)
    out['gg_batch_play_moves_tracked_T1_19x19_B65536'] = {'steps_per_s': B2 / t, 'note': 'one given move (a pass) per game per launch'}
    t = timed(lambda: gogame.batch_track(st2), 20)
    out['gg_batch_track_states_19x19_B65536'] = {'boards_per_s': B2 / t}
    t = timed(lambda: gogame.batch_untrack(tr), 20)
    out['gg_batch_untrack_states_19x19_B65536'] = {'boards_per_s': B2 / t}
    # policy-weighted sampling (gogame.random_weighted_action per game) and the env step that draws from weights itself
    probs = torch.rand((B2, N * N + 1), dtype=torch.float32, device='cuda')
    WB = 4 * (N * N + 1)
    t = timed(lambda: gogame.batch_sample_weighted(st2, probs, rt), 30)
    out['gg_batch_sample_weighted_19x19_B65536'] = {'boards_per_s': B2 / t, 'moved_GBps': B2 * (WB + N * N + 13) / t / 1e9}
    t = timed(lambda: gogame.batch_sample_weighted_rows(tr, N, probs, rt), 30)
    out['gg_batch_sample_weighted_rows_tracked_19x19_B65536'] = {'boards_per_s': B2 / t, 'moved_GBps': B2 * (WB + 4 * N + 16) / t / 1e9}
    obs = torch.empty_like(st2)
    ebuf = None

    def wstep(w):
        nonlocal ebuf
        ebuf = gogame.batch_env_step_tracked(tr, None, rt, 7.5, 'real', True, out=ebuf, states_out=obs, weights=w)
    t = timed(lambda: wstep(None), 30)
    t_w = timed(lambda: wstep(probs), 30)
    moved = 8 * (5 * N + 1) + S + 25
    out['gg_batch_env_step_tracked_19x19_B65536'] = {'steps_per_s': B2 / t, 'moved_GBps': B2 * moved / t / 1e9}
    out['gg_batch_env_step_tracked_weighted_19x19_B65536'] = {'steps_per_s': B2 / t_w, 'moved_GBps': B2 * (moved + WB) / t_w / 1e9,
                                                              'vs_uniform_draws': t / t_w}
    # batched symmetries: one view per game / all eight, byte planes and tracked boards
    orient = torch.randint(0, 8, (B2,), dtype=torch.int32, device='cuda')
    v1 = torch.empty_like(st2)
    t = timed(lambda: gogame.batch_symmetry(st2, orient, out=v1), 20)
    out['gg_batch_symmetry_one_view_19x19_B65536'] = {'boards_per_s': B2 / t, 'moved_GBps': B2 * 2 * S / t / 1e9}
    B3 = 8192
    v8 = torch.empty((B3, 8, 6, N, N), dtype=torch.uint8, device='cuda')
    t = timed(lambda: gogame.batch_symmetry(st2[:B3], None, out=v8), 20)
    out['gg_batch_symmetry_all_eight_19x19_B8192'] = {'boards_per_s': B3 / t, 'moved_GBps': B3 * 9 * S / t / 1e9}
    t = timed(lambda: gogame.batch_symmetry_rows(tr, N, orient), 20)
    out['gg_batch_symmetry_rows_tracked_19x19_B65536'] = {'boards_per_s': B2 / t, 'moved_GBps': B2 * 8 * (5 * N + 1) / t / 1e9}
    del probs, obs, v1, v8
    # replay of recorded move sequences (64 moves per game in one launch)
    T = 64
    rec = torch.empty((B2, T), dtype=torch.int32, device='cuda')
    la = torch.empty(B2, dtype=torch.int32, device='cuda')
    base = st2.clone()
    r3 = gogame.rng_seed(B2, 21)
    tmp = st2.clone()
    for tt in range(T):
        gogame.batch_rollout(tmp, r3, 1, False, la, None)
        rec[:, tt] = la
    work = base.clone()

    def replay():
        work.copy_(base)
        gogame.batch_play_moves(work, rec)
    t_copy = timed(lambda: work.copy_(base), 20)
    t = timed(replay, 10) - t_copy
    out['gg_batch_play_moves_T64_19x19_B65536'] = {'moves_per_s': B2 * float(gogame.batch_play_moves(base.clone(), rec).float().mean()) / t}
    # single-state latency (GoEnv.children / next_state path)
    one = st[0]
    t0 = time.perf_counter()
    for _ in range(20):
        gogame.children(one, canonical=True)
    torch.cuda.synchronize()
    out['children_single_state_19x19_ms'] = (time.perf_counter() - t0) / 20 * 1e3
    t0 = time.perf_counter()
    for _ in range(50):
        gogame.batch_next_states(st[:1], torch.tensor([361], device='cuda', dtype=torch.int32), check=False)
    torch.cuda.synchronize()
    out['next_state_single_19x19_ms'] = (time.perf_counter() - t0) / 50 * 1e3
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
