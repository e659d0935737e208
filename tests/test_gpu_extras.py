"""-m gpu: the helpers pinned by tests/golden/extras.npz (recorded from the reference by make_golden.py): gogame.str,
gogame.all_symmetries, state_utils.update_pieces on arbitrary positions; the GoEnv gym surface; the device guard."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def test_str_matches_reference_text(golden):
    """gogame.str (gym_go/gogame.py:407-468): the exact text the reference prints for 71 positions (2x2 ... 19x19,
    empty / middle game / passed / ended), NumPy and device-tensor inputs, and GoEnv.__str__."""
    from gymgo_amd import gogame
    from gymgo_amd.envs import GoEnv
    z = golden('extras')
    n = int(z['str/count'])
    assert n > 50
    for i in range(n):
        state, text = z['str/%d/state' % i], z['str/%d/text' % i].item()
        assert gogame.str(state.astype(np.float64)) == text, i
        if i % 7 == 0:
            assert gogame.str(torch.from_numpy(state).cuda()) == text, i
    state = z['str/5/state']
    env = GoEnv(size=state.shape[-1])
    env.state_ = state.astype(np.float64)
    assert str(env) == z['str/5/text'].item()


def test_all_symmetries_match_reference_views(golden):
    """gogame.all_symmetries (gym_go/gogame.py:362-382): the 8 views in the reference's order, for NumPy images and for
    device tensors (torch flips / rot90 on the device); random_symmetry returns one of them."""
    from gymgo_amd import gogame
    z = golden('extras')
    for j in range(int(z['sym/count'])):
        img, views = z['sym/%d/image' % j], z['sym/%d/views' % j]
        got = gogame.all_symmetries(img)
        got_dev = gogame.all_symmetries(torch.from_numpy(img).cuda())
        assert len(got) == len(got_dev) == 8
        for i in range(8):
            assert np.array_equal(np.asarray(got[i]), views[i]), (j, i)
            assert np.array_equal(got_dev[i].cpu().numpy(), views[i]), (j, i)
        r = gogame.random_symmetry(torch.from_numpy(img).cuda()).cpu().numpy()
        assert any(np.array_equal(r, v) for v in views)


def test_update_pieces_on_arbitrary_positions_matches_reference(golden):
    """state_utils.update_pieces (gym_go/state_utils.py:159-180) through gg_batch_update_pieces with the location lists
    as the reference receives them: 184 recorded cases on positions that are NOT reachable by legal play - all four
    corners, edges, duplicate locations, lists longer than four - states after, killed stones and group count."""
    from gymgo_amd import state_utils
    from oracle import c_oracle
    z = golden('extras')
    n = int(z['up/count'])
    hits = 0
    for i in range(n):
        k = 'up/%d/' % i
        state, adj, player = z[k + 'state'], z[k + 'adj'], int(z[k + 'player'])
        s = state.astype(np.float64)
        groups = state_utils.update_pieces(s, adj, player)
        assert np.array_equal(s.astype(np.uint8), z[k + 'after']), i
        killed = np.zeros(state.shape[1:], np.uint8)
        for g in groups:
            killed[g[:, 0], g[:, 1]] = 1
        assert np.array_equal(killed, z[k + 'killed']) and len(groups) == int(z[k + 'groups']), i
        hits += len(groups) > 0
        # the oracle agrees with the recorded reference output too (it is the checker of the batched test below)
        o_after, o_killed, o_n = c_oracle.update_pieces(state, adj[:, 0] * state.shape[-1] + adj[:, 1], player)
        assert np.array_equal(o_after, z[k + 'after']) and np.array_equal(o_killed, z[k + 'killed']) and o_n == len(groups)
    assert hits >= 25
    # batched form on a device tensor, K = 4, every corner of a 19x19 board: own stones on the corner AND on the
    # diagonal (the position that round 1's point-inference got wrong)
    N = 19
    batch = torch.zeros((8, 6, N, N), dtype=torch.uint8, device='cuda')
    adjs, players, want = [], [], []
    for b, (r, c) in enumerate([(0, 0), (0, N - 1), (N - 1, 0), (N - 1, N - 1)] * 2):
        dr, dc = (1 if r == 0 else -1), (1 if c == 0 else -1)
        pl = b // 4
        batch[b, pl, r, c] = 1
        batch[b, pl, r + dr, c + dc] = 1                      # the diagonal
        batch[b, 1 - pl, r + dr, c] = 1                       # two opponent stones next to the corner stone ...
        batch[b, 1 - pl, r, c + dc] = 1
        batch[b, pl, r + 2 * dr, c] = 1                       # ... each without a liberty
        batch[b, pl, r, c + 2 * dc] = 1
        adjs.append(np.array([[r + dr, c], [r, c + dc]]))
        players.append(pl)
    host = batch.cpu().numpy()
    for b in range(8):
        want.append(c_oracle.update_pieces(host[b], adjs[b][:, 0] * N + adjs[b][:, 1], players[b]))
    killed = state_utils.batch_update_pieces(np.arange(8), batch, adjs, players)
    for b in range(8):
        assert np.array_equal(batch[b].cpu().numpy(), want[b][0]), b
        assert len(killed[b]) == want[b][2] == 2 and int(batch[b, 1 - players[b]].sum()) == 0, b


def test_goenv_gym_surface():
    """gym_go/envs/go_env.py:19-37 + gym_go/__init__.py:3-10: a self-play loop written against the gym attributes
    only - observation_space / action_space (Box / Discrete, the real gym classes when gym is importable), sample(),
    contains - plays a legal game; registration happened on import when gym exists."""
    import gymgo_amd
    from gymgo_amd import envs
    from gymgo_amd.envs import spaces
    env = gymgo_amd.make('gym_go:go-v0', size=5, komi=0.5, reward_method='heuristic')
    assert isinstance(env, spaces.Env)
    assert env.action_space.n == 26 and env.observation_space.shape == (6, 5, 5)
    assert float(np.min(env.observation_space.low)) == 0.0 and float(np.max(env.observation_space.high)) == 6.0
    obs = env.reset()
    assert env.observation_space.contains(obs.astype(np.float32))
    assert envs.REGISTERED_WITH_GYM == (spaces.gym is not None)
    np.random.seed(20260927)          # (the draws below are NumPy's / the space's own: seeded, so the run is repeatable)
    if hasattr(env.action_space, 'seed'):
        env.action_space.seed(20260927)
    done, steps = False, 0
    while not done and steps < 300:
        a = env.action_space.sample()
        assert env.action_space.contains(a) and a in env.action_space
        if env.valid_moves()[a] == 0:
            with pytest.raises(AssertionError):
                env.step(a)
            a = env.uniform_random_action()
        obs, reward, done, info = env.step(a)
        assert obs.shape == env.observation_space.shape
        steps += 1
    assert steps >= 2                 # (a game ends with two passes in a row at the earliest)


def test_kernels_run_on_the_device_that_owns_the_buffers():
    """The C-ABI launches on the device of its buffer arguments and restores the caller's current device.  With one
    GPU visible this pins the guard's no-op path (current device unchanged, results correct from a side stream)."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    dev = torch.device('cuda', torch.cuda.device_count() - 1)
    before = torch.cuda.current_device()
    st = gogame.batch_init_state(300, 9, device=dev)
    rng = gogame.rng_seed(300, 5, 0, dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        gogame.batch_rollout(st, rng, 30, True)
    side.synchronize()
    assert torch.cuda.current_device() == before
    want, _, _ = c_oracle.batch_rollout(np.zeros((300, 6, 9, 9), np.uint8), c_oracle.rng_seed(5, 300), 30, True)
    assert np.array_equal(st.cpu().numpy(), want)
