"""-m gpu: policy-weighted action sampling (gogame.random_weighted_action / random_action, gym_go/gogame.py:385-404) and
batched symmetries (gogame.all_symmetries / random_symmetry, gym_go/gogame.py:340-382) on the device.

The weighted draw is defined in exact integers (include/gymgo_amd.h), so kernel == oracle is a bit-exact comparison;
the REFERENCE pins the distribution: its own normalised probabilities and 20 000 of its own seeded draws per weight
vector (tests/golden/policy.npz, made by tests/golden/make_golden.py from the imported reference)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _rng_t(rng_u64):
    return _dev(np.asarray(rng_u64, dtype=np.uint64).view(np.int64))


def _rng_np(rng_t):
    return rng_t.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize('N', [5, 9, 19])
def test_weighted_sampler_exact_golden_all_layouts(golden, N):
    """The committed (states, weights, generator) -> actions vectors of the exact sampler - states made by the reference,
    incl. a row without any playable weight (-1), a pass-only row, a negative weight, tiny (1e-30) rows - through the
    byte-plane kernel, the packed-row and the tracked-row kernels."""
    from gymgo_amd import gogame
    z = golden('policy')
    k = 'exact/%d/' % N
    states, w, rng0, rng1, want = (z[k + n] for n in ('states', 'weights', 'rng0', 'rng1', 'actions'))
    assert (want == -1).any() and (want == N * N).any() and (want >= 0).sum() > 50
    st, wt = _dev(states), _dev(w)
    for name, draw in (('bytes', lambda r: gogame.batch_sample_weighted(st, wt, r)),
                       ('packed', lambda r: gogame.batch_sample_weighted_rows(gogame.batch_pack(st), N, wt, r)),
                       ('tracked', lambda r: gogame.batch_sample_weighted_rows(gogame.batch_track(st), N, wt, r))):
        r = _rng_t(rng0)
        got = draw(r)
        assert np.array_equal(got.cpu().numpy(), want), name
        assert np.array_equal(_rng_np(r), rng1), name
    with pytest.raises(ValueError):
        gogame.batch_sample_weighted(st, wt, _rng_t(rng0), check=True)      # the all-zero row: the reference raises too
    with pytest.raises(ValueError):
        gogame.batch_sample_weighted(st, wt[:, :-1], _rng_t(rng0))


@pytest.mark.parametrize('N,B', [(2, 37), (7, 1000), (9, 4099), (13, 2050), (16, 777), (19, 8192)])
def test_weighted_sampler_vs_oracle(N, B):
    """Random policies on positions of every game phase (finished games included: nothing is masked there), weights
    spanning 60 orders of magnitude, zeros, negatives, NaN / inf (clamped on the bit pattern): kernel == oracle,
    generator states too; and without states (no mask)."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    gen = np.random.default_rng(N * 1000 + B)
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, 5 + N)
    for g in range(8):
        lo, hi = g * B // 8, (g + 1) * B // 8
        gogame.batch_rollout(st[lo:hi], rng[lo:hi], g * max(1, N * N // 4), False)
    A = N * N + 1
    w = (gen.random((B, A)) ** 4).astype(np.float32)
    w *= (10.0 ** gen.integers(-30, 30, size=(B, 1))).astype(np.float32)
    w[gen.random((B, A)) < 0.3] = 0
    w[::13, gen.integers(0, A)] = -1.0
    w[5::17, gen.integers(0, A)] = np.float32('nan')
    w[6::19, gen.integers(0, A)] = np.float32('inf')
    w[1] = 0
    host = st.cpu().numpy()
    assert host[:, 5, 0, 0].any()
    rng0 = _rng_np(rng).copy()
    want, rng1 = c_oracle.batch_sample_weighted(host, w, rng0)
    got = gogame.batch_sample_weighted(st, _dev(w), rng)
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(_rng_np(rng), rng1)
    assert want[1] == -1 and (want >= 0).mean() > 0.9
    want2, _ = c_oracle.batch_sample_weighted(None, w, rng0)
    got2 = gogame.batch_sample_weighted(None, _dev(w), _rng_t(rng0))
    assert np.array_equal(got2.cpu().numpy(), want2)
    assert (want2 != want).any() or N == 2
    # every drawn action is playable and carries weight
    ok = want >= 0
    inv = np.concatenate([host[:, 3].reshape(B, -1), np.zeros((B, 1), np.uint8)], axis=1)
    inv[host[:, 5, 0, 0] == 1] = 0
    assert not inv[ok, want[ok]].any()
    wv = w[ok, want[ok]]
    assert np.all(np.isnan(wv) | (wv > 0))


def test_weighted_sampler_distribution_matches_the_reference(golden):
    """20 000 device draws per weight vector against the probabilities the REFERENCE draws from (its own sklearn
    normalisation) and against 20 000 of the reference's own seeded draws: every action within 5 sigma of its expected
    count, chi-square of the two histograms in range.  Also gogame.random_action (weights 1 - invalid)."""
    from gymgo_amd import gogame
    z = golden('policy')
    n = 20000
    for c in range(int(z['case/count'])):
        k = 'case/%d/' % c
        N = int(z[k + 'size'])
        state, w, p, href = z[k + 'state'], z[k + 'weights'], z[k + 'p_reference'], z[k + 'hist_reference']
        st = _dev(state)[None].expand(n, -1, -1, -1).contiguous()
        wt = _dev(w)[None].expand(n, -1).contiguous()
        rng = gogame.rng_seed(n, 99 + c)
        got = gogame.batch_sample_weighted(st, wt, rng).cpu().numpy()
        assert (got >= 0).all()
        hist = np.bincount(got, minlength=N * N + 1)
        sigma = np.sqrt(n * p * (1 - p))
        assert np.all(np.abs(hist - n * p) <= 5 * sigma + 1.5), (c, str(z[k + 'kind']))
        assert not hist[p == 0].any()
        both = (hist + href) > 0
        chi2 = float((((hist - href) ** 2)[both] / (hist + href)[both]).sum())
        dof = int(both.sum()) - 1
        assert chi2 < dof + 6 * np.sqrt(2 * max(dof, 1)) + 10, (c, chi2, dof)
    for N in (2, 5, 9, 19):
        state, href = z['random_action/%d/state' % N], z['random_action/%d/hist_reference' % N]
        st = _dev(state)[None].expand(n, -1, -1, -1).contiguous()
        got = gogame.batch_random_action(st, gogame.rng_seed(n, 7 + N)).cpu().numpy()
        hist = np.bincount(got, minlength=N * N + 1)
        assert not hist[href == 0].any() and not href[hist == 0].any()
        both = (hist + href) > 0
        chi2 = float((((hist - href) ** 2)[both] / (hist + href)[both]).sum())
        dof = int(both.sum()) - 1
        assert chi2 < dof + 6 * np.sqrt(2 * max(dof, 1)) + 10


def _oracle_weighted_step(c_oracle, host, w, rng0, auto_reset):
    """GoEnv.step with a policy-weighted move per game, restated with the oracle's pieces: reset finished games
    (auto_reset), draw, refuse games that are frozen or have nothing to play, step the rest."""
    B = len(host)
    start = host.copy()
    over = start[:, 5, 0, 0] == 1
    frozen = over & (not auto_reset)
    if auto_reset:
        start[over] = 0
    draws = ~frozen
    acts = np.full(B, -1, np.int32)
    rng1 = rng0.copy()
    a, r = c_oracle.batch_sample_weighted(start[draws], w[draws], rng0[draws])
    acts[draws], rng1[draws] = a, r
    bad = acts < 0
    want = start.copy()
    ok = np.flatnonzero(~bad)
    nxt, status = c_oracle.batch_next_states(start[ok], acts[ok])
    assert not status.any()
    want[ok] = nxt
    return want, acts, bad.astype(np.int32), rng1


@pytest.mark.parametrize('N,B,auto_reset', [(9, 4099, True), (19, 8192, True), (19, 1000, False), (5, 16, True), (13, 70, True)])
def test_weighted_env_step_tracked_vs_oracle(N, B, auto_reset):
    """gg_batch_env_step_tracked_weighted: the move of every game drawn from policy weights INSIDE the step launch.
    States, observation, taken actions, status, dones, rewards and generator states against the oracle restatement,
    over several consecutive steps (so finished games, resets and refused all-zero rows occur)."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    gen = np.random.default_rng(N + B)
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, 11 + N)
    for g in range(8):
        lo, hi = g * B // 8, (g + 1) * B // 8
        gogame.batch_rollout(st[lo:hi], rng[lo:hi], g * max(1, N * N // 5), False)
    tracked = gogame.batch_track(st)
    obs = torch.empty_like(st)
    sd = torch.zeros(B, dtype=torch.int64, device='cuda')
    host = st.cpu().numpy()
    A = N * N + 1
    played = np.zeros(B, np.int64)
    seen_bad = seen_over = 0
    for step in range(6):
        w = (gen.random((B, A)) ** 6).astype(np.float32)
        w[:, -1] *= 30.0 if step >= 2 else 0.01          # later steps pass a lot: games end
        w[gen.random((B, A)) < 0.5] = 0
        w[step::29] = 0                                   # rows without any weight are refused
        rng0 = _rng_np(rng).copy()
        want, acts, status, rng1 = _oracle_weighted_step(c_oracle, host, w, rng0, auto_reset)
        seen_over += int((host[:, 5, 0, 0] == 1).sum())
        rewards, dones, st_out, taken = gogame.batch_env_step_tracked(tracked, None, rng, 0.5, 'real', auto_reset,
                                                                      states_out=obs, steps_done=sd, weights=_dev(w))
        assert np.array_equal(taken.cpu().numpy(), acts), step
        assert np.array_equal(st_out.cpu().numpy(), status), step
        assert np.array_equal(obs.cpu().numpy(), want), step
        assert np.array_equal(gogame.batch_untrack(tracked).cpu().numpy(), want), step
        assert np.array_equal(_rng_np(rng), rng1), step
        assert np.array_equal(dones.cpu().numpy(), want[:, 5, 0, 0]), step
        b_, w_ = c_oracle.batch_areas(want)
        margin = b_.astype(np.float64) - w_ - 0.5
        assert np.array_equal(rewards.cpu().numpy().astype(np.float64), np.where(want[:, 5, 0, 0] == 1, np.sign(margin), 0.0)), step
        played += (status == 0)
        seen_bad += int(status.sum())
        host = want
    assert np.array_equal(sd.cpu().numpy(), played)
    assert seen_bad > 0 and (seen_over > 0 or B < 100)
    with pytest.raises(ValueError):
        gogame.batch_env_step_tracked(tracked, torch.zeros(B, dtype=torch.int32, device='cuda'), rng, weights=_dev(w))
    with pytest.raises(ValueError):
        gogame.batch_env_step_tracked(tracked, None, None, weights=_dev(w))


@pytest.mark.parametrize('layout', ['tracked', 'bytes', 'packed'])
@pytest.mark.parametrize('auto_reset', [True, False])
def test_vecenv_step_with_probs_all_layouts_walk_the_same_games(layout, auto_reset):
    """GoVecEnv.step(probs=...) - fused with layout 'tracked', sample + step otherwise - against the oracle restatement;
    all three layouts therefore walk the same games.  Without auto_reset the finished games are frozen: they draw
    nothing (generator untouched, move -1, step refused) in every layout."""
    from gymgo_amd.envs import GoVecEnv
    from oracle import c_oracle
    B, N = 1500, 9
    env = GoVecEnv(B, N, komi=0.5, reward_method='real', seed=21, layout=layout, auto_reset=auto_reset)
    env.rollout(25)
    if not auto_reset:       # two passes in a row end the first 300 games: frozen from here on
        for _ in range(2):
            a = env.sample_actions()
            a[:300] = N * N
            env.step(a)
    host = env.states.cpu().numpy().copy()
    gen = np.random.default_rng(3)
    for step in range(5):
        w = (gen.random((B, N * N + 1)) ** 3).astype(np.float32)
        w[:, -1] *= 8.0
        rng0 = _rng_np(env.rng).copy()
        want, acts, status, rng1 = _oracle_weighted_step(c_oracle, host, w, rng0, auto_reset)
        if not auto_reset and step == 0:
            assert (host[:, 5, 0, 0] == 1).sum() > 10     # the case under test occurs
        obs, rewards, dones, st = env.step(probs=_dev(w))
        assert np.array_equal(obs.cpu().numpy() if layout != 'packed' else env.states.cpu().numpy(), want), (layout, step)
        assert np.array_equal(env.last_actions.cpu().numpy(), acts) and np.array_equal(st.cpu().numpy(), status)
        assert np.array_equal(_rng_np(env.rng), rng1)
        host = want
    with pytest.raises(ValueError):
        env.step(actions=torch.zeros(B, dtype=torch.int32, device='cuda'), probs=_dev(w))


# ------------------------------------------------------------------------------------------------ symmetries

def _np_view(img, o):
    """gym_go/gogame.py:373-381 restated with NumPy (checked against the reference's recorded views below)."""
    x = img
    if o & 1:
        x = np.flip(x, 2)
    if o & 2:
        x = np.flip(x, 1)
    if o & 4:
        x = np.rot90(x, axes=(1, 2))
    return np.ascontiguousarray(x)


def test_batch_symmetry_matches_reference_views(golden):
    """gg_batch_symmetry against the 8 views the REFERENCE's all_symmetries returned for arbitrary byte images
    (C = 6, 3, 1, 6; N = 5, 9, 19, 2): all eight at once, and one chosen view per image in a batch."""
    from gymgo_amd import gogame
    z = golden('extras')
    for j in range(int(z['sym/count'])):
        img, views = z['sym/%d/image' % j], z['sym/%d/views' % j]
        for o in range(8):
            assert np.array_equal(_np_view(img, o), views[o])          # the restatement used by the other tests
        batch = _dev(np.stack([img, img[:, ::-1].copy(), np.roll(img, 1, axis=2)]))
        all8 = gogame.batch_symmetry(batch).cpu().numpy()
        assert all8.shape == (3, 8) + img.shape
        assert np.array_equal(all8[0], views)
        for b in range(3):
            for o in range(8):
                assert np.array_equal(all8[b, o], _np_view(batch[b].cpu().numpy(), o))
        orient = np.array([5, 0, 3], np.int32)
        one = gogame.batch_symmetry(batch, orient).cpu().numpy()
        for b in range(3):
            assert np.array_equal(one[b], all8[b, orient[b]])


@pytest.mark.parametrize('N,C,B', [(19, 6, 301), (19, 1, 70), (9, 6, 515), (13, 3, 129), (7, 18, 65), (2, 6, 9), (19, 7, 40)])
def test_batch_symmetry_mixed_binary_and_arbitrary_games(N, C, B):
    """gg_batch_symmetry takes 0 / 1 planes through the bit domain and any other game of the SAME batch through the
    per-byte gather (decided per game, on the device): a batch that mixes both kinds - every third game carries bytes
    above 1, one of them only in its very last byte - on an UNALIGNED slice of a bigger buffer, all eight views and one
    random view per game, against the NumPy restatement of gym_go/gogame.py:373-381.  (19, 7): more rows than the bit
    path holds - the whole batch takes the generic kernel.)"""
    from gymgo_amd import gogame
    gen = np.random.default_rng(N * 100 + C)
    host = (gen.random((B, C, N, N)) < 0.4).astype(np.uint8)
    host[::3] = gen.integers(0, 256, size=host[::3].shape, dtype=np.uint8)
    host[1] = 0
    host[1, -1, -1, -1] = 200                       # arbitrary only in the last byte of the game
    host[2] = 1                                     # all ones: binary
    flat = torch.zeros(B * C * N * N + 64, dtype=torch.uint8, device='cuda')
    dev = flat[7:7 + B * C * N * N].view(B, C, N, N)          # 7 bytes off any alignment
    dev.copy_(_dev(host))
    out8 = torch.full((B * 8 * C * N * N + 64,), 255, dtype=torch.uint8, device='cuda')
    o8 = out8[3:3 + B * 8 * C * N * N].view(B, 8, C, N, N)
    all8 = gogame.batch_symmetry(dev, out=o8)
    got = all8.cpu().numpy()
    assert bool((out8[:3] == 255).all()) and bool((out8[3 + B * 8 * C * N * N:] == 255).all())   # nothing outside the range is touched
    for b in range(B):
        for o in range(8):
            assert np.array_equal(got[b, o], _np_view(host[b], o)), (b, o)
    orient = gen.integers(0, 8, size=B).astype(np.int32)
    one = gogame.batch_symmetry(dev, orient).cpu().numpy()
    for b in range(B):
        assert np.array_equal(one[b], got[b, orient[b]]), b


@pytest.mark.parametrize('N,B', [(5, 33), (9, 1001), (13, 257), (19, 4096)])
def test_symmetry_of_packed_and_tracked_boards_is_geometric(N, B):
    """Row-mask symmetries: pack(view(states)) == view_rows(pack(states)) and - because liberty classes and the invalid
    moves (ko included) are geometric - track(view(states)) == view_rows(track(states)), for a random orientation per
    game and for all eight.  Moves are equivariant: next_state(view(s), view_action(a)) == view(next_state(s, a)),
    checked with the oracle on the viewed side."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, 2 + N)
    for g in range(8):
        lo, hi = g * B // 8, (g + 1) * B // 8
        gogame.batch_rollout(st[lo:hi], rng[lo:hi], 3 + g * max(1, N * N // 5), False)
    host = st.cpu().numpy()
    views, orient = gogame.batch_random_symmetry(st, generator=torch.Generator(device='cuda').manual_seed(4))
    o = orient.cpu().numpy()
    assert len(np.unique(o)) == 8 or B < 64
    vh = views.cpu().numpy()
    for b in range(0, B, max(1, B // 64)):
        assert np.array_equal(vh[b], _np_view(host[b], int(o[b])))
    assert torch.equal(gogame.batch_symmetry_rows(gogame.batch_pack(st), N, orient), gogame.batch_pack(views))
    assert torch.equal(gogame.batch_symmetry_rows(gogame.batch_track(st), N, orient), gogame.batch_track(views))
    all8 = gogame.batch_symmetry(st)
    tr8 = gogame.batch_symmetry_rows(gogame.batch_track(st), N)
    pk8 = gogame.batch_symmetry_rows(gogame.batch_pack(st), N)
    for v in range(8):
        assert torch.equal(tr8[:, v].contiguous(), gogame.batch_track(all8[:, v].contiguous())), v
        assert torch.equal(pk8[:, v].contiguous(), gogame.batch_pack(all8[:, v].contiguous())), v
    # equivariance of moves
    live = host[:, 5, 0, 0] == 0
    acts = gogame.batch_sample_actions(st, rng)
    nxt, _ = gogame.batch_next_states(st, acts, check=False)     # (finished games: refused, passed through)
    va = gogame.symmetry_actions(acts, orient, N)
    want, status = c_oracle.batch_next_states(vh, va.cpu().numpy())
    assert not status[live].any()
    got = gogame.batch_symmetry(nxt, orient).cpu().numpy()
    assert np.array_equal(got[live], want[live])
    with pytest.raises(ValueError):
        gogame.batch_symmetry_rows(gogame.batch_pack(st), N + 1, orient)


@pytest.mark.parametrize('dtype', ['bfloat16', 'float16'])
def test_half_precision_weights_draw_like_their_float32_values(dtype):
    """bfloat16 / float16 policy weights (GG_W_BF16 / GG_W_F16) are widened exactly by the kernels: the draw is the
    oracle's draw on the same values as float32 - stand-alone on byte planes / packed / tracked boards and fused into
    the tracked env step; half the bytes read."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    tdt = getattr(torch, dtype)
    B, N = 3000, 19
    gen = np.random.default_rng(5)
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, 12)
    for g in range(8):
        lo, hi = g * B // 8, (g + 1) * B // 8
        gogame.batch_rollout(st[lo:hi], rng[lo:hi], g * 45, False)
    w32 = (gen.random((B, N * N + 1)) ** 5).astype(np.float32)
    w32 *= (10.0 ** gen.integers(-4, 3, size=(B, 1))).astype(np.float32)
    w32[gen.random(w32.shape) < 0.4] = 0
    w32[7] = 0
    wh = torch.from_numpy(w32).cuda().to(tdt)                  # what a bf16 / fp16 policy head hands over
    exact = wh.float().cpu().numpy()                           # ... and its exact float32 values
    assert (exact != w32).any()
    host = st.cpu().numpy()
    rng0 = _rng_np(rng).copy()
    want, rng1 = c_oracle.batch_sample_weighted(host, exact, rng0)
    for draw in (lambda r: gogame.batch_sample_weighted(st, wh, r),
                 lambda r: gogame.batch_sample_weighted_rows(gogame.batch_pack(st), N, wh, r),
                 lambda r: gogame.batch_sample_weighted_rows(gogame.batch_track(st), N, wh, r)):
        r = _rng_t(rng0)
        assert np.array_equal(draw(r).cpu().numpy(), want)
        assert np.array_equal(_rng_np(r), rng1)
    assert want[7] == -1 and (want >= 0).sum() > B - 50
    # fused into the env step
    tracked = gogame.batch_track(st)
    obs = torch.empty_like(st)
    r = _rng_t(rng0)
    w_want, w_acts, w_status, w_rng1 = _oracle_weighted_step(c_oracle, host, exact, rng0, True)
    rewards, dones, status, taken = gogame.batch_env_step_tracked(tracked, None, r, 0.0, 'real', True, states_out=obs, weights=wh)
    assert np.array_equal(taken.cpu().numpy(), w_acts) and np.array_equal(status.cpu().numpy(), w_status)
    assert np.array_equal(obs.cpu().numpy(), w_want) and np.array_equal(_rng_np(r), w_rng1)
    # float64 weights are converted to float32 by the wrapper
    got64 = gogame.batch_sample_weighted(st, torch.from_numpy(exact.astype(np.float64)).cuda(), _rng_t(rng0))
    assert np.array_equal(got64.cpu().numpy(), want)
