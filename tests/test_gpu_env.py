"""-m gpu: GoEnv / GoVecEnv on the device against the golden env traces recorded from the reference's GoEnv."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def _mv(m):
    return tuple(m) if isinstance(m, list) else m


def test_goenv_scripted_traces(golden, scripted_cases):
    """States, rewards, done flags and info dicts of every scripted sequence, through GoEnv.step like the
    reference's own tests drive it (tuple / int / None actions)."""
    from gymgo_amd.envs import make
    z = golden('scripted')
    for c in scripted_cases:
        n = c['name']
        env = make('gym_go:go-v0', size=c['size'], komi=c.get('komi', 0), reward_method=c.get('reward_method', 'real'))
        s0 = env.reset()
        assert s0.shape == (6, c['size'], c['size']) and s0.dtype == np.float64 and not s0.any()
        moves = list(c['moves']) + c.get('continue', [])
        for i, m in enumerate(moves):
            if i == len(c['moves']):
                for bad in c.get('then_raises', []):
                    with pytest.raises(Exception):
                        env.step(_mv(bad))
            state, reward, done, info = env.step(_mv(m))
            assert np.array_equal(state.astype(np.uint8), z[n + '/states'][i]), (n, i)
            assert float(reward) == float(z[n + '/rewards'][i]), (n, i)
            assert int(done) == int(z[n + '/dones'][i])
            assert info['turn'] == int(z[n + '/turns'][i])
            assert np.array_equal(np.asarray(info['invalid_moves']).astype(np.uint8), z[n + '/invalid_moves'][i])
            assert bool(info['prev_player_passed']) == bool(z[n + '/prev_passed'][i])
            assert list(env.gogame.num_liberties(state)) == list(z[n + '/num_liberties'][i])
        if len(moves) == len(c['moves']):
            for bad in c.get('then_raises', []):
                with pytest.raises(Exception):
                    env.step(_mv(bad))


def test_goenv_misc_api():
    from gymgo_amd import gogame
    from gymgo_amd.envs import GoEnv
    env = GoEnv(size=7)
    with pytest.raises(Exception):
        env.step((-1, 0))          # test_invalid_moves.py:19-24
    with pytest.raises(Exception):
        env.step((0, 100))
    s = env.state()
    nxt = gogame.next_state(s, 0)  # input untouched (test_basics.py:48-52)
    assert not s.any() and nxt[0, 0, 0] == 1
    for fmt in ((1, 2), [1, 2], np.array([1, 2]), 9):   # test_basics.py:70-81
        env.reset()
        st, _, _, _ = env.step(fmt)
        assert st[0, 1, 2] == 1
    env.reset()
    np.random.seed(20260927)       # (uniform_random_action draws with NumPy's global generator, like the reference)
    for _ in range(20):
        env.step(env.uniform_random_action())
        if env.game_ended():
            break
    for canon in (False, True):    # test_basics.py:209-223
        kids = env.children(canonical=canon, padded=True)
        vm = env.valid_moves()
        for a in range(50):
            if vm[a]:
                assert np.array_equal(kids[a], gogame.next_state(env.state(), a, canon))
            else:
                assert not kids[a].any()
    assert 'Turn:' in str(env)
    assert env.canonical_state()[2].max() == 0
    assert len(gogame.children(env.state(), padded=False)) == int(env.valid_moves().sum())


LAYOUTS = ['tracked', 'bytes']


@pytest.mark.parametrize('layout', LAYOUTS)
def test_vecenv_step_and_rollout_agree_with_oracle(layout):
    from gymgo_amd.envs import GoVecEnv
    from oracle import c_oracle
    B, N = 300, 9
    env = GoVecEnv(B, N, komi=5.5, reward_method='real', seed=77, layout=layout)
    want = np.zeros((B, 6, N, N), np.uint8)
    rng = c_oracle.rng_seed(77, B)
    for t in range(120):
        acts = env.sample_actions()
        states, rewards, dones, status = env.step(acts)
        ended = want[:, 5, 0, 0] == 1
        want[ended] = 0
        want, rng, last = c_oracle.batch_rollout(want, rng, 1, False)
        assert np.array_equal(acts.cpu().numpy(), last)
        assert np.array_equal(states.cpu().numpy(), want), t
        b, w = c_oracle.batch_areas(want)
        ref_r = np.where(want[:, 5, 0, 0] == 1, np.sign(b - w - 5.5), 0.0)
        assert np.array_equal(rewards.cpu().numpy(), ref_r)
        assert np.array_equal(dones.cpu().numpy(), want[:, 5, 0, 0])
        assert int(status.sum()) == 0
    env.rollout(33)
    want, rng, _ = c_oracle.batch_rollout(want, rng, 33, True)
    assert np.array_equal(env.states.cpu().numpy(), want)
    assert int(env.steps_done.min()) == 153


def _ref_reward(states, komi, method, N):
    from oracle import c_oracle
    b, w = c_oracle.batch_areas(states)
    margin = b.astype(np.float64) - w - komi
    over = states[:, 5, 0, 0] == 1
    if method == 'real':
        return np.where(over, np.sign(margin), 0.0)
    return np.where(over, np.where(margin > 0, 1.0, -1.0) * N * N, margin)


@pytest.mark.parametrize('layout', LAYOUTS)
@pytest.mark.parametrize('N,B,method,komi', [(19, 257, 'heuristic', 7.5), (9, 300, 'real', 0.0), (5, 64, 'real', 2.0),
                                             (13, 33, 'heuristic', 0.0), (2, 17, 'real', 0.0), (9, 5000, 'real', 6.5)])
def test_fused_env_step_sampled_matches_oracle(N, B, method, komi, layout):
    """gg_batch_env_step / gg_batch_env_step_tracked with on-device sampling: states (the observation the step
    returns), drawn actions, dones and GoEnv.reward vs the oracle (gym_go/envs/go_env.py:49-76, :128-149) over many
    plies incl. game ends and auto-resets."""
    from gymgo_amd.envs import GoVecEnv
    from oracle import c_oracle
    env = GoVecEnv(B, N, komi=komi, reward_method=method, seed=4242, layout=layout)
    want = np.zeros((B, 6, N, N), np.uint8)
    rng = c_oracle.rng_seed(4242, B)
    seen_done = 0
    for t in range(40 if N == 19 else 150):
        states, rewards, dones, status = env.step()
        want, rng, last = c_oracle.batch_rollout(want, rng, 1, True)
        assert np.array_equal(env.last_actions.cpu().numpy(), last), t
        assert np.array_equal(states.cpu().numpy(), want), t
        assert np.array_equal(dones.cpu().numpy(), want[:, 5, 0, 0])
        assert np.array_equal(rewards.cpu().numpy().astype(np.float64), _ref_reward(want, komi, method, N)), t
        assert int(status.sum()) == 0
        seen_done += int(dones.sum())
    assert np.array_equal(env.rng.cpu().numpy().view(np.uint64), rng)
    if N <= 9:
        assert seen_done > 0   # the reset + terminal-reward branches were exercised


@pytest.mark.parametrize('layout', LAYOUTS)
def test_fused_env_step_refuses_illegal_and_frozen(layout):
    from gymgo_amd import gogame
    from gymgo_amd.envs import GoVecEnv
    from oracle import c_oracle
    B, N = 203, 9
    env = GoVecEnv(B, N, komi=0.5, reward_method='heuristic', seed=5, auto_reset=False, layout=layout)
    env.rollout(40)
    before = env.states.clone()
    host = before.cpu().numpy()
    gen = np.random.default_rng(0)
    acts = gen.integers(-3, N * N + 4, size=B).astype(np.int32)
    over = host[:, 5, 0, 0] == 1
    inval = host[:, 3].reshape(B, -1)
    in_range = (acts >= 0) & (acts <= N * N)
    bad = ~in_range | over
    for i in range(B):
        if in_range[i] and acts[i] < N * N and inval[i, acts[i]]:
            bad[i] = True
    assert bad.any() and (~bad).any()
    n0 = env.steps_done.clone()
    states, rewards, dones, status = env.step(torch.from_numpy(acts).cuda())
    assert np.array_equal((env.steps_done - n0).cpu().numpy(), (~bad).astype(np.int64))   # refused steps are not counted
    want = host.copy()
    ok = np.flatnonzero(~bad)
    want[ok] = c_oracle.batch_next_states(host[ok], acts[ok])[0]
    assert np.array_equal(status.cpu().numpy(), bad.astype(np.int32))
    assert np.array_equal(states.cpu().numpy(), want)
    assert np.array_equal(dones.cpu().numpy(), want[:, 5, 0, 0])
    assert np.array_equal(rewards.cpu().numpy().astype(np.float64), _ref_reward(want, 0.5, 'heuristic', N))
    # the separate-launch form of the same step agrees
    env2 = GoVecEnv(B, N, komi=0.5, reward_method='heuristic', seed=5, auto_reset=False, layout='bytes')
    env2.states = before.clone()
    legal = torch.from_numpy(np.where(bad, N * N, acts).astype(np.int32)).cuda()
    s2, r2, d2, st2 = env2.step_unfused(legal)
    live = torch.from_numpy(~over).cuda()
    env3 = GoVecEnv(B, N, komi=0.5, reward_method='heuristic', seed=5, auto_reset=False, layout=layout)
    env3.states = before.clone()
    s3, r3, d3, st3 = env3.step(legal)
    assert torch.equal(s2[live], s3[live]) and torch.equal(r2[live].float(), r3[live])
    # null outputs / bad arguments
    with pytest.raises(ValueError):
        gogame.batch_env_step(env.states)
    with pytest.raises(KeyError):
        gogame.batch_env_step(env.states, legal, reward_method='nope')
    with pytest.raises(ValueError):
        gogame.batch_env_step_tracked(gogame.batch_track(env.states))
    # an auto-resetting env resets a finished game even when the move given for it is then refused (GoEnv.reset comes
    # before the action check), and a reset game accepts any in-range move
    env4 = GoVecEnv(B, N, komi=0.5, reward_method='real', seed=5, auto_reset=True, layout=layout)
    env4.states = before.clone()
    acts4 = acts.copy()
    acts4[np.flatnonzero(over)[::2]] = -1               # half of the finished games get an out-of-range move
    s4, r4, d4, st4 = env4.step(torch.from_numpy(acts4).cuda())
    start = host.copy()
    start[over] = 0
    inval4 = start[:, 3].reshape(B, -1)
    in_range = (acts4 >= 0) & (acts4 <= N * N)
    bad4 = ~in_range
    for i in range(B):
        if in_range[i] and acts4[i] < N * N and inval4[i, acts4[i]]:
            bad4[i] = True
    want4 = start.copy()
    ok4 = np.flatnonzero(~bad4)
    want4[ok4] = c_oracle.batch_next_states(start[ok4], acts4[ok4])[0]
    assert np.array_equal(st4.cpu().numpy(), bad4.astype(np.int32))
    assert np.array_equal(s4.cpu().numpy(), want4)
    assert over.any() and (bad4 & over).any()


@pytest.mark.parametrize('layout', LAYOUTS)
def test_vecenv_step_captured_in_hipgraph(layout):
    """GoVecEnv.step() allocates nothing (fixed output buffers), so K steps with on-device sampling can be captured
    in a hipGraph; a replay walks the same trajectory as the fused rollout and leaves the last step's rewards."""
    from gymgo_amd import gogame
    from gymgo_amd.envs import GoVecEnv
    B, N, K = 2048, 9, 10
    env = GoVecEnv(B, N, komi=0.5, reward_method='heuristic', seed=3, layout=layout)
    env.rollout(25)
    s0, r0, n0 = env.states.clone(), env.rng.clone(), env.steps_done.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        env.step()
    torch.cuda.current_stream().wait_stream(side)
    env.states = s0.clone(); env.rng.copy_(r0); env.steps_done.copy_(n0)
    states_buf = env.states
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(K):
            states, rewards, dones, status = env.step()
    if layout == 'bytes':
        states_buf.copy_(s0)
    else:
        env.states = s0.clone()
    env.rng.copy_(r0); env.steps_done.copy_(n0)
    graph.replay()
    torch.cuda.synchronize()
    want, wr = s0.clone(), r0.clone()
    gogame.batch_rollout(want, wr, K, True)
    assert torch.equal(env.states, want) and torch.equal(env.rng, wr)
    assert int((env.steps_done - n0).min()) == K and int(status.sum()) == 0
    b, w = gogame.batch_areas(want)
    margin = (b - w).float() - 0.5
    over = want[:, 5, 0, 0] == 1
    assert torch.equal(rewards, torch.where(over, torch.where(margin > 0, 81.0, -81.0), margin))
    assert torch.equal(dones, want[:, 5, 0, 0])


def test_symmetries_and_helpers_on_device_tensors():
    """all_symmetries / random_symmetry (gym_go/gogame.py:338-382: h-flip bit 0, v-flip bit 1, rot90 bit 2) on device
    tensors against the same NumPy operations; liberties / num_liberties / areas helpers on a played position."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    N = 7
    st = gogame.batch_init_state(1, N, device='cuda')
    rng = gogame.rng_seed(1, 12)
    gogame.batch_rollout(st, rng, 23, False)
    img = st[0]
    host = img.cpu().numpy()
    syms = gogame.all_symmetries(img)
    assert len(syms) == 8
    for i, x in enumerate(syms):
        want = host
        if (i >> 0) % 2:
            want = np.flip(want, 2)
        if (i >> 1) % 2:
            want = np.flip(want, 1)
        if (i >> 2) % 2:
            want = np.rot90(want, axes=(1, 2))
        assert np.array_equal(x.cpu().numpy(), want), i
    r = gogame.random_symmetry(img).cpu().numpy()
    assert any(np.array_equal(r, s.cpu().numpy()) for s in syms)
    # areas of every symmetry equal the areas of the position (scoring is symmetric)
    b0, w0 = gogame.areas(img)
    for x in syms:
        assert gogame.areas(x.contiguous()) == (b0, w0)
    ob, ow = c_oracle.batch_areas(host[None])
    assert (int(b0), int(w0)) == (int(ob[0]), int(ow[0]))
    libs_b, libs_w = gogame.liberties(host)
    nb, nw = gogame.num_liberties(host)
    assert int(libs_b.sum()) == int(nb) and int(libs_w.sum()) == int(nw)
    empties = (host[0] + host[1]) == 0
    assert not (libs_b & ~empties).any() and not (libs_w & ~empties).any()


def _cat_step(outs):
    return [torch.cat([o[i] for o in outs]) for i in range(4)]


@pytest.mark.parametrize('layout,B,parts', [('tracked', 4096, 2), ('tracked', 1001, 3), ('bytes', 515, 2), ('packed', 700, 4)])
def test_vecenv_parts_walk_the_same_games_as_one_env(layout, B, parts):
    """GoVecEnvParts: the sub-batches step on streams of their own, in lock-step (step) or each at its own pace
    (step_part / wait), and are the games of ONE GoVecEnv - same boards, generators, rewards, move counts - with
    device-drawn moves, given moves and policy weights."""
    from gymgo_amd.envs import GoVecEnv, GoVecEnvParts
    N = 9
    kw = dict(komi=0.5, reward_method='heuristic', seed=11, layout=layout)
    one, many = GoVecEnv(B, N, **kw), GoVecEnvParts(B, N, parts=parts, **kw)
    assert [hi - lo for lo, hi in many.bounds] == [e.batch_size for e in many.envs] and many.bounds[-1][1] == B
    one.rollout(30)
    for h in range(parts):
        many.rollout_part(h, 30)
    assert torch.equal(many.gather('states'), one.states) and torch.equal(many.gather('rng'), one.rng)
    g = torch.Generator(device='cuda').manual_seed(5)
    for k in range(6):
        if k % 3 == 0:
            kwargs = {}
        elif k % 3 == 1:
            kwargs = {'probs': torch.rand((B, N * N + 1), device='cuda', generator=g) ** 4}
        else:
            kwargs = {'actions': one.sample_actions().clone()}   # (drawn from a COPY of the generators below)
            one.rng.copy_(many.gather('rng'))
        want = [t.clone() for t in one.step(**kwargs)]
        got = _cat_step(many.step(**kwargs))
        for w, t in zip(want, got):
            assert torch.equal(w, t)
        assert torch.equal(many.gather('rng'), one.rng) and torch.equal(many.gather('last_actions'), one.last_actions)
    assert torch.equal(many.gather('steps_done'), one.steps_done)
    # each part at its own pace: three steps of part 0 before part 1 has made any
    order = [h for h in range(parts) for _ in range(3)]
    for h in order:
        many.step_part(h)
    for _ in range(3):
        one.step()
    assert torch.equal(many.gather('states'), one.states) and torch.equal(many.gather('rng'), one.rng)
    assert torch.equal(many.gather('steps_done'), one.steps_done)
    last = [many.wait(h) for h in range(parts)]
    torch.cuda.synchronize()
    assert all(many.ready(h) for h in range(parts))
    assert torch.equal(torch.cat([o[1] for o in last]), one._step_out[0])


def test_vecenv_parts_step_captured_in_hipgraph():
    """The parts' streams fork from and join the capturing stream: K lock-step rounds of a two-part env replay as one
    hipGraph and land where the fused rollout lands."""
    from gymgo_amd import gogame
    from gymgo_amd.envs import GoVecEnvParts
    B, N, K = 4096, 9, 8
    env = GoVecEnvParts(B, N, parts=2, komi=0.5, seed=3)
    for h in range(2):
        env.rollout_part(h, 25)
    s0, r0 = env.gather('states').clone(), env.gather('rng').clone()
    env.step()                                 # warm-up outside the capture
    torch.cuda.synchronize()

    def restore():
        for e, (lo, hi) in zip(env.envs, env.bounds):
            e.states = s0[lo:hi].clone(); e.rng.copy_(r0[lo:hi])
    restore()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(K):
            outs = env.step()
    restore()
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    want, wr = s0.clone(), r0.clone()
    gogame.batch_rollout(want, wr, K, True)
    assert torch.equal(env.gather('states'), want) and torch.equal(env.gather('rng'), wr)
    assert int(torch.cat([o[3] for o in outs]).sum()) == 0


def test_vecenv_step_follows_rebound_buffers_and_refuses_bad_ones():
    """GoVecEnv.step() (tracked layout) keeps the validated pointers of the env's buffers between steps: re-binding a buffer
    (generator states, tracked boards, step counters, komi / reward method) is picked up by the next step - two envs, one of
    them re-bound to clones and changed settings between steps, walk the same games - and a buffer of the wrong type is
    refused exactly as gogame.batch_env_step_tracked refuses it."""
    from gymgo_amd import _lib, gogame
    from gymgo_amd.envs import GoVecEnv
    a = GoVecEnv(777, 9, komi=0.5, reward_method='real', seed=5)
    b = GoVecEnv(777, 9, komi=0.5, reward_method='real', seed=5)
    for i in range(60):
        if i % 7 == 3:
            b.rng = b.rng.clone()
            b.tracked = b.tracked.clone()
            b.steps_done = b.steps_done.clone()
        if i == 30:
            a.komi = b.komi = 6.5
            a.reward_method = b.reward_method = 'heuristic'
        acts = None if i % 3 else gogame.batch_sample_actions(a.states, gogame.rng_seed(777, 1000 + i))   # (a throw-away generator)
        oa, ra, da, sa = a.step(acts)
        ob, rb, db, sb = b.step(None if acts is None else acts.clone())
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(sa, sb), i
        assert torch.equal(a.rng, b.rng) and torch.equal(a.tracked, b.tracked) and torch.equal(a.steps_done, b.steps_done)
    assert int(a.steps_done.min()) > 0
    b.rng = b.rng.to(torch.float64)
    with pytest.raises(_lib.GymGoNativeError):
        b.step()
    b.rng = a.rng.clone()
    b.tracked = b.tracked[:-1]
    with pytest.raises(ValueError):
        b.step()


@pytest.mark.parametrize('N,B,method,komi', [(7, 1, 'real', 0.0), (9, 300, 'heuristic', 6.5), (19, 5000, 'real', 7.5), (13, 33, 'real', 0.5),
                                             (5, 4099, 'heuristic', 0.0)])
def test_env_step_scored_areas_rewards_and_states_match_oracle(N, B, method, komi):
    """gg_batch_env_step_scored (round 6: GoEnv.step in ONE launch): the step of gg_batch_env_step plus the Tromp-Taylor areas
    (gym_go/gogame.py:275-300) of every resulting position - states, drawn actions, dones, rewards by either formula and the
    areas vs the oracle, over game ends and auto-resets, on both sides of the four-wave-workgroup threshold."""
    from gymgo_amd import _lib, gogame
    from oracle import c_oracle
    L = _lib.lib()
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, 99, 0, 'cuda')
    want = np.zeros((B, 6, N, N), np.uint8)
    wrng = c_oracle.rng_seed(99, B)
    rew = torch.empty(B, dtype=torch.float32, device='cuda')
    dones = torch.empty(B, dtype=torch.uint8, device='cuda')
    status = torch.empty(B, dtype=torch.int32, device='cuda')
    taken = torch.empty(B, dtype=torch.int32, device='cuda')
    areas = torch.empty((B, 2), dtype=torch.int32, device='cuda')
    m = 0 if method == 'real' else 1
    seen_done = 0
    for t in range(12 if N == 19 else 120):
        assert L.gg_batch_env_step_scored(st.data_ptr(), None, rng.data_ptr(), rew.data_ptr(), dones.data_ptr(), status.data_ptr(),
                                          taken.data_ptr(), areas.data_ptr(), B, N, komi, m, 1, None) == 0
        want, wrng, last = c_oracle.batch_rollout(want, wrng, 1, True)
        assert np.array_equal(st.cpu().numpy(), want), t
        assert np.array_equal(taken.cpu().numpy(), last), t
        b, w = c_oracle.batch_areas(want)
        assert np.array_equal(areas.cpu().numpy(), np.stack([b, w], axis=1)), t
        assert np.array_equal(rew.cpu().numpy().astype(np.float64), _ref_reward(want, komi, method, N)), t
        assert np.array_equal(dones.cpu().numpy(), want[:, 5, 0, 0]) and int(status.sum()) == 0
        seen_done += int(dones.sum())
    assert np.array_equal(rng.cpu().numpy().view(np.uint64), wrng)
    if N <= 9 and B > 1:
        assert seen_done > 0
    # given actions incl. an illegal one: status 1, the row and its areas are those of the untouched position
    acts = torch.full((B,), N * N, dtype=torch.int32, device='cuda')
    occupied = np.flatnonzero(want[0, 3].reshape(-1))
    if len(occupied):
        acts[0] = int(occupied[0])
    before = st.clone()
    assert L.gg_batch_env_step_scored(st.data_ptr(), acts.data_ptr(), None, None, dones.data_ptr(), status.data_ptr(), None,
                                      areas.data_ptr(), B, N, komi, m, 1, None) == 0
    if len(occupied) and not want[0, 5, 0, 0]:
        assert int(status[0]) == 1 and torch.equal(st[0], before[0])
        b0, w0 = c_oracle.batch_areas(want[:1])
        assert areas[0].tolist() == [int(b0[0]), int(w0[0])]
    assert L.gg_batch_env_step_scored(st.data_ptr(), None, None, None, None, None, None, areas.data_ptr(), B, N, komi, m, 1, None) == -2
    assert L.gg_batch_env_step_scored(st.data_ptr(), acts.data_ptr(), None, None, None, None, None, None, B, N, komi, m, 1, None) == -2
    assert L.gg_batch_env_step_scored(st.data_ptr(), acts.data_ptr(), None, None, None, None, None, areas.data_ptr(), B, N, komi, 5, 1, None) == -3


def test_goenv_step_is_one_launch_on_a_pinned_record():
    """GoEnv.step (config 1): the game's record lives in pinned, device-mapped host memory; one step = one launch of
    gg_batch_env_step_scored and a stream wait - a game replayed against the oracle incl. captures, the end of the game, a
    caller who replaces / edits `state_` between steps, and an illegal move that leaves everything untouched."""
    from gymgo_amd.envs import make
    from oracle import c_oracle
    env = make('gym_go:go-v0', size=7, komi=0.5, reward_method='heuristic')
    state = env.reset()
    want = np.zeros((6, 7, 7), np.uint8)
    rs = np.random.default_rng(3)
    for t in range(400):
        if env.done:
            break
        valid = np.flatnonzero(env.valid_moves())
        a = int(rs.choice(valid))
        state, reward, done, info = env.step(a)
        want = c_oracle.next_state(want, a)
        assert np.array_equal(state, want.astype(np.float64)), t
        b, w = c_oracle.batch_areas(want[None])
        margin = float(b[0]) - float(w[0]) - 0.5
        assert reward == ((1 if margin > 0 else -1) * 49 if done else margin), t
        assert done == int(want[5, 0, 0]) and info['turn'] == int(want[2, 0, 0])
    assert env._dev is not None and env._dev['buf'].is_pinned() and not env._dev['buf'].is_cuda
    # an illegal move: AssertionError, nothing changes
    env.reset()
    env.step(8)
    snap = env.state()
    with pytest.raises(AssertionError):
        env.step(8)
    assert np.array_equal(env.state(), snap) and not env.done
    # the caller edits state_ in place: the next step starts from the edited position (gym_go recomputes from state_)
    env.state_[0, 3, 3] = 1
    edited = env.state().astype(np.uint8)
    s2, *_ = env.step(0)
    assert np.array_equal(s2, c_oracle.next_state(edited, 0).astype(np.float64))
