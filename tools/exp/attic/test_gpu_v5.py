"""-m gpu: the 32-boards-per-wave per-ply kernels (gymgo_amd/csrc/gg_v5.h) against the oracle at the batch sizes that
dispatch to them (full-size boards, >= CUs x 128 boards), tails that do not fill a wave, every kind of move."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def _positions(gogame, B, N, seed):
    """Positions from every game phase (no auto-reset: finished games stay on the board) + a mixed bag of moves."""
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, seed)
    for g in range(16):
        lo, hi = g * B // 16, (g + 1) * B // 16
        gogame.batch_rollout(st[lo:hi], rng[lo:hi], g * max(1, (N * N) // 9), False)
    acts = gogame.batch_sample_actions(st, rng).cpu().numpy().copy()
    gen = np.random.default_rng(seed)
    pick = gen.random(B)
    acts[pick < 0.10] = N * N                                           # passes
    rnd = gen.integers(0, N * N, size=B)
    acts[(pick >= 0.10) & (pick < 0.25)] = rnd[(pick >= 0.10) & (pick < 0.25)]   # arbitrary points: many are illegal
    acts[(pick >= 0.25) & (pick < 0.27)] = N * N + 1                    # out of range
    acts[(pick >= 0.27) & (pick < 0.28)] = -2
    return st, acts.astype(np.int32)


@pytest.mark.parametrize('N,canonical', [(19, False), (19, True), (13, False), (9, True)])
def test_next_states_32_boards_per_wave_vs_oracle(N, canonical):
    """gg_batch_next_states at 32 805 boards (1 025 full waves + a 5-board tail): states and status against the oracle for
    every board - captures, kos, passes, game ends, occupied / suicide / ko points and out-of-range moves (rows pass
    through) - and against the two-boards-per-wave kernel on slices (an independent HIP path)."""
    from gymgo_amd import gogame, _lib
    from oracle import c_oracle
    cus = _lib.lib().gg_device_cus()
    B = cus * 128 + 37
    st, acts = _positions(gogame, B, N, 40 + N)
    host = st.cpu().numpy()
    want, wstat = c_oracle.batch_next_states_mt(host, acts, canonical)
    bad = wstat != 0
    want[bad] = host[bad]
    got, stat = gogame.batch_next_states(st, torch.from_numpy(acts).cuda(), canonical, check=False)
    assert np.array_equal(stat.cpu().numpy(), wstat)
    assert np.array_equal(got.cpu().numpy(), want)
    assert bad.sum() > B // 50 and (acts == N * N).sum() > B // 20
    assert (host[:, 5, 0, 0] == 1).any()                       # finished games: next_state plays on (gogame.py:34-87)
    # captures and kos did occur
    stones_in = host[:, 0].sum(axis=(1, 2)) + host[:, 1].sum(axis=(1, 2))
    stones_out = want[:, 0].sum(axis=(1, 2)) + want[:, 1].sum(axis=(1, 2))
    assert (stones_out < stones_in).sum() > B // 100
    # the small-batch kernel agrees (slices below the dispatch threshold)
    for lo in range(0, B, 8192):
        g2, s2 = gogame.batch_next_states(st[lo:lo + 8192], torch.from_numpy(acts[lo:lo + 8192]).cuda(), canonical, check=False)
        assert torch.equal(g2, got[lo:lo + 8192]) and torch.equal(s2, stat[lo:lo + 8192])
    assert np.array_equal(st.cpu().numpy(), host)              # the input batch is untouched


def test_next_states_32_rollout_loop_vs_oracle():
    """A rollout through the step API at 65 536 boards x 19x19: sample + next_states for 40 plies (every output is the
    next input), a strided sub-sample replayed by the oracle."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    B, N = 65536, 19
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, 3)
    gogame.batch_rollout(st, rng, 150, True)
    idx = np.arange(7, B, 64)
    idx_t = torch.as_tensor(idx, device='cuda')
    want = st[idx_t].cpu().numpy()
    for ply in range(40):
        gogame.batch_reset_finished(st)
        acts = gogame.batch_sample_actions(st, rng)
        st = gogame.batch_next_states(st, acts)
        w0 = want.copy()
        w0[w0[:, 5, 0, 0] == 1] = 0
        want, ws = c_oracle.batch_next_states_mt(w0, acts[idx_t].cpu().numpy())
        assert not ws.any()
        assert np.array_equal(st[idx_t].cpu().numpy(), want), ply
