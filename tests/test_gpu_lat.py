"""The latency-shaped multi-ply kernel (gymgo_amd/csrc/gg_lat.h, k_rollout_lat) where gg_batch_rollout dispatches it - and on
both sides of every take-over point - against the pinned C oracle: states, generator states, last actions and step counters.

gg_batch_rollout serves a launch from one of three kernel families by (board size, games, plies per launch)
(gg_kernels.hip: use_lat, use_multi_ply, the per-ply kernels); the result must not depend on which.  Reference loop:
gym_go/envs/go_env.py:49-81 over gym_go/gogame.py:34-87.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cus():
    from gymgo_amd import _lib
    return int(_lib.lib().gg_device_cus())


def _run(N, B, launches, auto_reset, seed=77, first_game=0):
    """`launches` plies-per-launch in a row on one batch; every launch compared with the oracle."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, seed, first_game, 'cuda')
    want = np.zeros((B, 6, N, N), np.uint8)
    want_rng = rng.cpu().numpy().view(np.uint64).copy()
    sd = torch.zeros(B, dtype=torch.int64, device='cuda')
    played = np.zeros(B, np.int64)
    for F in launches:
        la = torch.full((B,), -9, dtype=torch.int32, device='cuda')
        gogame.batch_rollout(st, rng, F, auto_reset, la, sd)
        want, want_rng, want_last = c_oracle.batch_rollout_mt(want, want_rng, F, auto_reset)
        got = st.cpu().numpy()
        bad = np.flatnonzero((got != want).reshape(B, -1).any(axis=1))
        assert len(bad) == 0, (N, B, F, auto_reset, bad[:6].tolist())
        assert np.array_equal(rng.cpu().numpy().view(np.uint64), want_rng), (N, B, F)
        assert np.array_equal(la.cpu().numpy(), want_last), (N, B, F)
        played += F
    if auto_reset:      # every game plays every ply; a frozen game (auto_reset off) plays none after its second pass
        assert np.array_equal(sd.cpu().numpy(), played)
    else:
        assert bool((sd.cpu() <= torch.from_numpy(played)).all()) and int(sd.min()) > 0
    return st


# (board size, games): every row capacity (9 / 13 / 19 and sizes below them), whole and ragged waves (4 boards per wave up to
# 13x13, 2 at 19x19), one board, and the config-2 batch
@pytest.mark.parametrize('N,B', [(9, 4096), (9, 1), (9, 2), (9, 3), (9, 5), (9, 1001), (7, 130), (5, 77), (2, 9), (3, 10), (8, 64),
                                 (13, 1023), (13, 6), (11, 100), (12, 67), (10, 33),
                                 (19, 511), (19, 2), (19, 3), (19, 1), (16, 33), (14, 40), (15, 17)])
def test_lat_rollout_vs_oracle(N, B):
    """Launch lengths from the shortest the kernel takes (3 / 4 / 64 plies) to whole games with auto-reset, and frozen
    games (auto_reset off: a finished game neither moves nor draws)."""
    launches = (3, 4, 64, 65, 3 * N * N + 9) if N <= 13 else (64, 70, 3 * N * N + 9)
    if B > 1000:
        launches = launches[:-1] + (200,)
    _run(N, B, launches, True)
    _run(N, B, launches[-2:] + launches[:1], False, seed=5)


@pytest.mark.parametrize('N', [9, 13, 19, 7])
def test_rollout_same_result_on_both_sides_of_every_take_over(N):
    """The games-per-launch and plies-per-launch thresholds of use_lat (and of use_multi_ply above them): the batch sizes /
    launch lengths right at, below and above each take-over point, every launch against the oracle."""
    cus = _cus()
    per_cu, min_plies = (128, 3) if N <= 9 else (80, 3) if N <= 13 else (31, 8)     # (64 / 32 / 8 games per CU until round 6)
    edge = cus * per_cu
    for B in (edge - 3, edge, edge + 1, edge + 5):
        _run(N, B, (min_plies - 1, min_plies, min_plies + 1, 40 if N <= 13 else 90), True, seed=B)
    if 9 < N <= 13:   # (13x13: from 4 plies per launch below 16 games per CU, from 3 above)
        for B in (cus * 16 - 1, cus * 16):
            _run(N, B, (2, 3, 4, 5), True, seed=B)
    # below / above the multi-ply kernel's own take-over (32 games per CU), launches the new kernel does not take
    for B in (cus * 32 - 2, cus * 32 + 2):
        _run(N, B, (1, 2), True, seed=B + 1)


def test_lat_unaligned_views_and_neighbours_untouched():
    """Boards at odd byte offsets (views into a larger buffer): results equal the aligned run, bytes outside the batch keep
    their value (the emitter writes ragged edges as single bytes)."""
    from gymgo_amd import gogame
    for N, B in ((9, 37), (13, 21), (19, 9)):
        S = 6 * N * N
        for off in (1, 7, 16, 33):
            buf = torch.full((off + B * S + 64,), 0xAB, dtype=torch.uint8, device='cuda')
            view = buf[off:off + B * S].view(B, 6, N, N)
            view.zero_()
            ref = gogame.batch_init_state(B, N, device='cuda')
            r1, r2 = gogame.rng_seed(B, 9, 0, 'cuda'), gogame.rng_seed(B, 9, 0, 'cuda')
            for F in (70, 120):
                gogame.batch_rollout(view, r1, F, True)
                gogame.batch_rollout(ref, r2, F, True)
            assert torch.equal(view, ref) and torch.equal(r1, r2), (N, off)
            assert bool((buf[:off] == 0xAB).all()) and bool((buf[off + B * S:] == 0xAB).all()), (N, off)


def test_lat_starts_from_arbitrary_midgame_positions():
    """The first classes of a launch come from the stones alone (eleven lock-step floods): start launches from positions
    produced by OTHER kernels (per-ply path, big-batch path) at many depths."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    for N, B in ((9, 512), (13, 256), (19, 128)):
        st = gogame.batch_init_state(B, N, device='cuda')
        rng = gogame.rng_seed(B, 31, 0, 'cuda')
        want = np.zeros((B, 6, N, N), np.uint8)
        want_rng = rng.cpu().numpy().view(np.uint64).copy()
        for depth in (1, 2, 1, 2, 1):          # one- and two-ply launches: the per-ply kernels
            gogame.batch_rollout(st, rng, depth, True)
            want, want_rng, _ = c_oracle.batch_rollout_mt(want, want_rng, depth, True)
        for F in (64, 1, 80, 2, 150, 1, 64):   # alternate the new kernel with per-ply launches
            gogame.batch_rollout(st, rng, F, True)
            want, want_rng, _ = c_oracle.batch_rollout_mt(want, want_rng, F, True)
            assert np.array_equal(st.cpu().numpy(), want), (N, F)
            assert np.array_equal(rng.cpu().numpy().view(np.uint64), want_rng)


@pytest.mark.parametrize('N,B', [(9, 4096), (9, 5), (9, 1001), (7, 130), (5, 77), (2, 9), (13, 1023), (11, 100), (19, 511), (19, 3), (16, 33)])
def test_lat_tracked_rollout_vs_oracle(N, B):
    """gg_batch_rollout_tracked on batches the latency-shaped kernel takes (tracked boards: from ONE ply per launch on): the
    untracked boards == the oracle, the tracked words == gg_batch_track_states of the oracle's boards bit for bit (liberty
    classes are a function of the position), generator states and last actions as well; auto-reset on and off."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    for auto_reset in (True, False):
        st = gogame.batch_init_state(B, N, device='cuda')
        tr = gogame.batch_track(st)
        rng = gogame.rng_seed(B, 11 + N, 0, 'cuda')
        want, want_rng = st.cpu().numpy(), rng.cpu().numpy().view(np.uint64).copy()
        sd = torch.zeros(B, dtype=torch.int64, device='cuda')
        for F in (1, 1, 2, 7, 64, 3 * N * N + 5):
            la = torch.full((B,), -9, dtype=torch.int32, device='cuda')
            gogame.batch_rollout_tracked(tr, rng, F, auto_reset, la, sd)
            want, want_rng, want_last = c_oracle.batch_rollout_mt(want, want_rng, F, auto_reset)
            assert np.array_equal(gogame.batch_untrack(tr).cpu().numpy(), want), (N, B, F, auto_reset)
            assert torch.equal(gogame.batch_track(torch.from_numpy(want).cuda()), tr), (N, B, F, auto_reset)
            assert np.array_equal(rng.cpu().numpy().view(np.uint64), want_rng) and np.array_equal(la.cpu().numpy(), want_last)
        if auto_reset:
            assert int(sd.min()) == int(sd.max()) == 1 + 1 + 2 + 7 + 64 + 3 * N * N + 5


@pytest.mark.parametrize('N', [9, 13, 19])
def test_tracked_rollout_same_result_on_both_sides_of_the_take_over(N):
    """64 / 64 / 16 games per CU: the batch sizes right at and above the point where k_rollout4 takes tracked launches back."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    edge = _cus() * (64 if N <= 13 else 16)
    for B in (edge, edge + 1):
        st = gogame.batch_init_state(B, N, device='cuda')
        tr = gogame.batch_track(st)
        rng = gogame.rng_seed(B, B, 0, 'cuda')
        want, want_rng = st.cpu().numpy(), rng.cpu().numpy().view(np.uint64).copy()
        for F in (1, 3, 40):
            gogame.batch_rollout_tracked(tr, rng, F, True)
            want, want_rng, _ = c_oracle.batch_rollout_mt(want, want_rng, F, True)
            assert np.array_equal(gogame.batch_untrack(tr).cpu().numpy(), want), (N, B, F)
            assert np.array_equal(rng.cpu().numpy().view(np.uint64), want_rng)


def _env_step_vs_oracle(N, B, with_obs, given):
    """One gg_batch_env_step_tracked launch on mid-game boards (some finished: auto-reset) against one oracle ply."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, 3 * N + B % 7, 0, 'cuda')
    q = max(1, B // 4)
    for g in range(4):
        lo, hi = g * q, (B if g == 3 else (g + 1) * q)
        if lo < hi:
            gogame.batch_rollout(st[lo:hi], rng[lo:hi], 5 + (N * N * g) // 3, False)
    want, want_rng = st.cpu().numpy(), rng.cpu().numpy().view(np.uint64).copy()
    tr = gogame.batch_track(st)
    obs = torch.empty_like(st) if with_obs else None
    sd = torch.zeros(B, dtype=torch.int64, device='cuda')
    want2, rng2, last2 = c_oracle.batch_rollout_mt(want, want_rng, 1, True)
    acts = torch.from_numpy(last2).cuda() if given else None        # (the oracle's own moves, handed over as given actions)
    r_dev = rng.clone()
    rewards, dones, status, taken = gogame.batch_env_step_tracked(tr, acts, None if given else r_dev, 0.5, 'real', True,
                                                                  states_out=obs, steps_done=sd)
    assert np.array_equal(gogame.batch_untrack(tr).cpu().numpy(), want2), (N, B, with_obs, given)
    if with_obs:
        assert np.array_equal(obs.cpu().numpy(), want2)
    assert int(status.abs().sum()) == 0 and np.array_equal(taken.cpu().numpy(), last2) and int(sd.min()) == int(sd.max()) == 1
    if not given:
        assert np.array_equal(r_dev.cpu().numpy().view(np.uint64), rng2)
    over = want2[:, 5, 0, 0] == 1
    assert np.array_equal(dones.cpu().numpy().astype(bool), over)
    b, w = c_oracle.batch_areas_mt(want2)
    assert np.array_equal(rewards.cpu().numpy().astype(np.float64), np.where(over, np.sign(b.astype(np.float64) - w - 0.5), 0.0))


@pytest.mark.parametrize('N,B', [(9, 4096), (9, 5), (13, 1001), (19, 511), (19, 3), (5, 77), (16, 33)])
def test_lat_env_step_vs_oracle(N, B):
    """GoVecEnv.step's launch on the latency-shaped kernel: drawn and given moves, with and without the observation
    (tests/test_gpu_env.py drives the same entry point through GoVecEnv: refused / frozen / reset cases, heuristic reward)."""
    for with_obs in (True, False):
        for given in (False, True):
            _env_step_vs_oracle(N, B, with_obs, given)


def test_env_step_same_result_on_both_sides_of_the_19x19_take_over():
    """19x19: k_env_step_lat up to 128 games per CU with the observation (64 without), k_rollout4's env instantiation above."""
    cus = _cus()
    for per_cu, with_obs in ((128, True), (64, False)):
        for B in (cus * per_cu, cus * per_cu + 1):
            _env_step_vs_oracle(19, B, with_obs, False)


@pytest.mark.parametrize('N', [9, 13, 19, 6])
def test_four_wave_workgroups_ragged_and_at_their_take_over(N):
    """Short launches go out as FOUR-wave workgroups (k_rollout2_w4 on byte planes up to 16 / 8 pairs per CU; k_rollout_lat_w4 /
    k_env_step_lat_w4 on tracked boards up to 13x13 and 16 groups of four per CU, <= 4 plies): batches that leave the last
    workgroup with one, two or three waves (and the last wave ragged), and the batch sizes on both sides of each take-over -
    rollouts, tracked rollouts and env steps against the oracle."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    cus = _cus()
    edge_bytes = 2 * cus * (16 if N <= 9 else 8)          # games = 2 x pairs
    edge_tracked = 4 * cus * 16
    for B in (1, 2, 3, 7, 8, 9, 13, 17, 31, 33, edge_bytes - 1, edge_bytes, edge_bytes + 1, edge_bytes + 2):
        _run(N, B, (1, 2, 1), True, seed=B + 3)
    for B in (1, 4, 5, 9, 15, 16, 17, 29, edge_tracked - 1, edge_tracked, edge_tracked + 1):
        st = gogame.batch_init_state(B, N, device='cuda')
        tr = gogame.batch_track(st)
        rng = gogame.rng_seed(B, B + 11, 0, 'cuda')
        want, want_rng = st.cpu().numpy(), rng.cpu().numpy().view(np.uint64).copy()
        for F in (1, 4, 5, 2, 33, 1):
            gogame.batch_rollout_tracked(tr, rng, F, True)
            want, want_rng, _ = c_oracle.batch_rollout_mt(want, want_rng, F, True)
            assert np.array_equal(gogame.batch_untrack(tr).cpu().numpy(), want), (N, B, F)
            assert np.array_equal(rng.cpu().numpy().view(np.uint64), want_rng)
    for B in (1, 5, 13, 17, edge_tracked, edge_tracked + 1):
        for with_obs, given in ((True, False), (False, True)):
            _env_step_vs_oracle(N, B, with_obs, given)


@pytest.mark.parametrize('N', [9, 13, 19, 6])
def test_three_flood_ply_second_round_on_a_point_between_three_groups(N):
    """Round 6: the usual ply of the one-row-per-lane kernels floods THREE sets (two opponent slots + the mover's group) and
    runs a second round when the new stone touches a THIRD distinct opponent group with >= 2 liberties (four are impossible:
    a point whose four neighbours are opponent stones is only playable when one of them is in atari).  Crafted positions put
    an empty point between three separate white groups of different shapes - every choice of the fourth, empty, side, so that
    the left-over neighbour of round two is the left or the right one - on every board of a batch; the boards differ only in
    their generators, so a known share of them plays that point on the first ply.  Byte planes (first classes from scratch)
    and tracked boards, long launches (the three-flood ply) and short ones (the five-flood ply), all against the oracle."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    B = 768 if N <= 13 else 4096
    c = N // 2
    sides = ((c - 1, c), (c + 1, c), (c, c - 1), (c, c + 1))          # up, down, left, right of (c, c)
    for skip in range(4):
        s0 = np.zeros((6, N, N), np.uint8)
        for i, (r, q) in enumerate(sides):
            if i != skip:
                s0[1, r, q] = 1                     # three white stones around (c, c), pairwise not adjacent
        if N >= 9:                                  # different shapes: the upper one grows upwards, the right one to the right
            if skip != 0:
                s0[1, c - 2, c] = s0[1, c - 3, c] = 1
            if skip != 3:
                s0[1, c, c + 2] = 1
            s0[0, 0, 0] = s0[0, N - 1, N - 1] = 1   # black stones elsewhere
        s0[3] = c_oracle.compute_invalid_moves(s0, 0)
        assert s0[3, c, c] == 0                     # the point between the three groups is playable for black
        states = np.repeat(s0[None], B, axis=0)
        for tracked in (False, True):
            for launches in ((40, 3, 64), (1, 2, 1, 50)):
                st = torch.from_numpy(states).cuda()
                rng = gogame.rng_seed(B, 1234 + N + skip, 0, 'cuda')
                want, want_rng = states.copy(), rng.cpu().numpy().view(np.uint64).copy()
                _, _, last1 = c_oracle.batch_rollout_mt(want, want_rng, 1, True)
                assert int((last1 == c * N + c).sum()) >= 3     # some boards play that point on the first ply
                tr = gogame.batch_track(st) if tracked else None
                for F in launches:
                    if tracked:
                        gogame.batch_rollout_tracked(tr, rng, F, True)
                    else:
                        gogame.batch_rollout(st, rng, F, True)
                    want, want_rng, _ = c_oracle.batch_rollout_mt(want, want_rng, F, True)
                    got = gogame.batch_untrack(tr).cpu().numpy() if tracked else st.cpu().numpy()
                    assert np.array_equal(got, want), (N, skip, tracked, F)
                    assert np.array_equal(rng.cpu().numpy().view(np.uint64), want_rng), (N, skip, tracked, F)
