"""-m gpu: the headline kernel oracle-checked WHERE THE BENCH RUNS IT.

  * bench.run_rank itself (HipBackend, 65 536 x 19x19, default de-synchronisation / burn-in / warm-up / 4 timed launches
    of 256 plies) with a strided sub-sample of the games replayed by the C oracle after EVERY launch: thousand-ply
    stationary boards, game ends, auto-resets and re-openings included (gym_go/gogame.py:34-87, gym_go/envs/go_env.py:78-81);
  * the multi-ply kernel at its own dispatch size for >= 2 000 plies (several whole games per slot);
  * gg_batch_invalid_mask with a ko point and an explicit player (gym_go/state_utils.py:24-83), which the other GPU
    tests only ever call with ko=None, player=None.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def _seeds(c_oracle, base, idx):
    return np.array([c_oracle.lib().gg_oracle_rng_seed(int(base), int(i)) for i in idx], dtype=np.uint64)


def _checked_backend(sample):
    """bench.HipBackend whose every launch is followed by an oracle replay of `sample` games chosen by index inside the
    rank's shard (generators seeded by GLOBAL game index): states, generator states and, at the end, the kernel's own
    step counters must agree."""
    import bench
    from oracle import c_oracle

    class Checked(bench.HipBackend):
        launches = 0
        deepest = 0

        def setup(self, count, size, first_game):
            super().setup(count, size, first_game)
            self.idx = np.arange(5, count, count // sample)
            self.idx_t = torch.as_tensor(self.idx, device=self.device)
            self.want = np.zeros((len(self.idx), 6, size, size), np.uint8)
            self.want_rng = _seeds(c_oracle, bench.SEED, first_game + self.idx)
            self.plies = np.zeros(len(self.idx), np.int64)       # plies played by each sub-sample game so far
            self.counted = np.zeros(len(self.idx), np.int64)     # ... of which the launch was asked to count

        def rollout(self, plies, lo=0, hi=None, count_steps=True):
            hi = self.count if hi is None else hi
            super().rollout(plies, lo, hi, count_steps)
            sel = np.flatnonzero((self.idx >= lo) & (self.idx < hi))
            if plies <= 0 or len(sel) == 0:
                return
            w, r, _ = c_oracle.batch_rollout_mt(self.want[sel], self.want_rng[sel], plies, True)
            self.want[sel], self.want_rng[sel] = w, r
            self.plies[sel] += plies
            if count_steps and lo == 0 and hi == self.count:
                self.counted[sel] += plies
            sel_t = torch.as_tensor(self.idx[sel], device=self.device)
            assert np.array_equal(self.states[sel_t].cpu().numpy(), w), ('states', self.launches, plies, lo, hi)
            assert np.array_equal(self.rng[sel_t].cpu().numpy().view(np.uint64), r), ('rng', self.launches)
            self.launches += 1
            self.deepest = int(self.plies.max())

    return Checked(torch.device('cuda', 0))


def test_bench_trajectory_subsample_vs_oracle():
    """The bench's own driver on the bench's own workload: every launch bench.run_rank issues (15 de-synchronising
    slice launches of 40 ... 600 plies, 1 burn-in + 5 warm-up + 4 timed launches of 256 plies over all 65 536 games -
    k_rollout4<19, 0, false, true, false, false>) is followed by an oracle replay of 512 games chosen by global game index
    (every 128th, offset 5: all 16 slices); states, generator states and the kernel's own step counters must agree
    after each of the 25 launches (+ the untimed clock-settle launches of the same shape), i.e. up to ~3 160 plies into a slot's life (mean game length ~640)."""
    import bench
    back = _checked_backend(512)
    opts = {'size': 19, 'plies_per_step': 256, 'steps': 4, 'warmup': 5, 'games_per_gpu': 65536, 'desync': 640,
            'burn_in_steps': 1, 'world': 1}
    res = bench.run_rank(0, 1, back, opts, None)
    assert res['steps_played'] == 4 * 256 * 65536
    assert back.launches == 15 + 1 + res['settle_launches'] + 5 + 4 and res['settle_launches'] >= 4     # (+ the clock-settle launches, untimed)
    assert back.deepest >= 2560 + 560 + 256 * res['settle_launches']
    assert np.array_equal(back.steps_done[back.idx_t].cpu().numpy(), back.counted)
    # the sub-sample really is the stationary mix: dense boards, finished-and-restarted games, every slice
    stones = back.want[:, 0].sum(axis=(1, 2)) + back.want[:, 1].sum(axis=(1, 2))
    assert stones.mean() > 150 and stones.min() < 80 and stones.max() > 280
    assert len(np.unique(back.idx // 4096)) == 16
    # and the whole batch agrees with its own sub-sample replay in aggregate (no game was skipped)
    assert int(back.steps_done.min()) == 4 * 256 and int(back.steps_done.max()) == 4 * 256


def test_bench_driver_as_rank_5_of_8_at_the_per_gpu_size_vs_oracle():
    """BASELINE config 4 as the N > 1 bench lines run it: bench.run_rank itself as RANK 5 OF 8 - 131 072 games per GPU,
    the shard's first global game 655 360, the default de-synchronisation / burn-in / warm-up and 3 timed launches of
    256 plies (two rounds of waves of k_rollout4 per launch) - with 512 games of the shard replayed by the oracle after
    every launch.  (No process group: the rank is driven on its own; the two-rank reductions are tests/test_multirank_gloo.py.)"""
    import bench
    back = _checked_backend(512)
    opts = {'size': 19, 'plies_per_step': 256, 'steps': 3, 'warmup': 2, 'games_per_gpu': 131072, 'desync': 640,
            'burn_in_steps': 1, 'world': 8}
    res = bench.run_rank(5, 8, back, opts, None)
    assert (res['first'], res['count'], res['total_games']) == (5 * 131072, 131072, 1048576)
    assert res['per_rank'] == [dict(res['per_rank'][0], first_game=655360, steps_played=3 * 256 * 131072)]
    assert back.launches == 15 + 1 + res['settle_launches'] + 2 + 3 and back.deepest >= 6 * 256 + 560
    assert np.array_equal(back.steps_done[back.idx_t].cpu().numpy(), back.counted)
    assert int(back.steps_done.min()) == 3 * 256 and int(back.steps_done.max()) == 3 * 256
    assert len(np.unique(back.idx // 8192)) == 16       # every de-synchronisation slice is in the sub-sample


@pytest.mark.parametrize('B,launch', [(16384, 256), (8192, 173)])
def test_multi_ply_kernel_deep_run_vs_oracle(B, launch):
    """k_rollout4 at its own dispatch sizes (16 384 games = 16 boards per wave, 8 192 = 8 per wave) for >= 2 048 plies
    at 19x19 - three to four whole games per slot incl. their ends, resets and re-openings; every 16th game replayed by
    the oracle after each launch (states, generator, last action), step counters at the end."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    N, seed, total = 19, 77, 2048
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, seed)
    la = torch.empty(B, dtype=torch.int32, device='cuda')
    sd = torch.zeros(B, dtype=torch.int64, device='cuda')
    idx = np.arange(0, B, 16)
    idx_t = torch.as_tensor(idx, device='cuda')
    want = np.zeros((len(idx), 6, N, N), np.uint8)
    want_rng = _seeds(c_oracle, seed, idx)
    done, restarts = 0, 0
    stones = np.zeros(len(idx), np.int64)
    while done < total:
        k = min(launch, total - done)
        gogame.batch_rollout(st, rng, k, True, la, sd)
        want, want_rng, want_last = c_oracle.batch_rollout_mt(want, want_rng, k, True)
        done += k
        assert np.array_equal(st[idx_t].cpu().numpy(), want), ('states', done)
        assert np.array_equal(rng[idx_t].cpu().numpy().view(np.uint64), want_rng), ('rng', done)
        assert np.array_equal(la[idx_t].cpu().numpy(), want_last), ('last action', done)
        now = want[:, 0].sum(axis=(1, 2)).astype(np.int64) + want[:, 1].sum(axis=(1, 2))
        restarts += int((now < stones - 100).sum())
        stones = now
    assert int(sd.min()) == total and int(sd.max()) == total
    assert restarts > len(idx)      # the games ended and re-opened (more than once per slot on average)


@pytest.mark.parametrize('N', [2, 3, 5, 7, 9, 10, 13, 16, 19])
def test_invalid_mask_with_ko_and_explicit_player_vs_oracle(N):
    """state_utils.batch_compute_invalid_moves(batch_state, batch_player, batch_ko_protect) -> gg_batch_invalid_mask
    with a ko pointer, for BOTH players on every position whatever its turn plane says, and with ko points on empty
    and on occupied points / no ko; oracle = gg_oracle_compute_invalid_moves(state, player, ko)
    (gym_go/state_utils.py:24-83: the mask is for the OPPONENT of `player`)."""
    from gymgo_amd import gogame, state_utils
    from oracle import c_oracle
    B = 384 if N > 9 else 256
    gen = np.random.default_rng(100 + N)
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, 9 + N)
    for g in range(8):          # game phases from the opening to late boards (no auto-reset: finished games stay full)
        lo, hi = g * B // 8, (g + 1) * B // 8
        gogame.batch_rollout(st[lo:hi], rng[lo:hi], 1 + g * max(2, N * N // 6), False)
    host = st.cpu().numpy()
    players = gen.integers(0, 2, size=B).astype(np.int64)
    ko = gen.integers(-1, N * N, size=B).astype(np.int32)
    ko[::3] = -1
    want = np.stack([c_oracle.compute_invalid_moves(host[i], int(players[i]), int(ko[i])) for i in range(B)])
    # int tensor of flat indices
    got = state_utils.batch_compute_invalid_moves(st, players, torch.from_numpy(ko).cuda())
    assert np.array_equal((got.cpu().numpy() > 0).astype(np.uint8), want)
    # list of None / (r, c), numpy states in (the reference's own argument forms), the other player
    ko_list = [None if k < 0 else (int(k) // N, int(k) % N) for k in ko]
    want2 = np.stack([c_oracle.compute_invalid_moves(host[i], 1 - int(players[i]), int(ko[i])) for i in range(B)])
    got2 = state_utils.batch_compute_invalid_moves(host.astype(np.float64), 1 - players, ko_list)
    assert got2.dtype == np.bool_ and np.array_equal(got2.astype(np.uint8), want2)
    assert (want != want2).any() or N == 2     # the player argument matters
    # player=None: the turn plane decides (player = 1 - turn), ko still applied
    turn = host[:, 2, 0, 0].astype(np.int64)
    want3 = np.stack([c_oracle.compute_invalid_moves(host[i], 1 - int(turn[i]), int(ko[i])) for i in range(B)])
    got3 = state_utils.batch_compute_invalid_moves(st, None, torch.from_numpy(ko).cuda())
    assert np.array_equal((got3.cpu().numpy() > 0).astype(np.uint8), want3)
    # single-state form
    i = int(np.flatnonzero(ko >= 0)[0])
    one = state_utils.compute_invalid_moves(host[i].astype(np.float64), int(players[i]), ko_list[i])
    assert np.array_equal(one.astype(np.uint8), want[i])
    assert np.array_equal(st.cpu().numpy(), host)     # inputs untouched (the turn plane is rewritten on a copy)


@pytest.mark.parametrize('B', [2, 16, 33])
def test_tracked_env_step_resets_finished_games_whose_move_is_refused(B):
    """EVERY game of the batch (hence of each wave) is finished, auto_reset is on and every action is out
    of range: the games are reset (GoEnv.reset precedes the action check, gym_go/envs/go_env.py:40-57) and the moves
    refused - exactly what the byte-plane env step does."""
    from gymgo_amd import gogame
    N = 9
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, 3)
    gogame.batch_rollout(st, rng, 30, False)
    passes = torch.full((B,), N * N, dtype=torch.int32, device='cuda')
    st = gogame.batch_next_states(st, passes)
    st = gogame.batch_next_states(st, passes)
    assert bool((st[:, 5, 0, 0] == 1).all()) and int(st[:, 0].sum()) > 0
    bad = torch.full((B,), N * N + 1, dtype=torch.int32, device='cuda')
    bad[B // 2:] = -7
    ref = st.clone()
    r_ref = gogame.batch_env_step(ref, bad, None, 0.0, 'real', True)
    tracked = gogame.batch_track(st)
    obs = torch.empty_like(st)
    r_trk = gogame.batch_env_step_tracked(tracked, bad, None, 0.0, 'real', True, states_out=obs)
    assert not ref.any()                                   # reset boards
    assert torch.equal(gogame.batch_untrack(tracked), ref)
    assert torch.equal(obs, ref)
    for a, b in zip(r_ref, r_trk):                         # rewards, dones, status (all refused), actions
        assert torch.equal(a, b)
    assert bool((r_trk[2] == 1).all()) and not bool(r_trk[1].any())
