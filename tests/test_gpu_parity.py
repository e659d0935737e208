"""-m gpu: the HIP path (through the C-ABI) against the oracle and the golden vectors. Bit-exact."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def gg():
    assert torch.cuda.is_available(), 'gpu tests need a ROCm device'
    from gymgo_amd import gogame
    return gogame


@pytest.fixture(scope='module')
def oracle():
    from oracle import c_oracle
    c_oracle.lib()
    return c_oracle


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def h64(a):
    return np.frombuffer(hashlib.blake2b(np.ascontiguousarray(a).tobytes(), digest_size=8).digest(), dtype=np.uint64)[0]


def test_scripted_golden(gg, golden, scripted_cases):
    """Every scripted sequence of the reference's unit tests: per-step states equal the reference's."""
    z = golden('scripted')
    for c in scripted_cases:
        n, size = c['name'], c['size']
        acts, states = z[n + '/actions'], z[n + '/states']
        s = torch.zeros((6, size, size), dtype=torch.uint8, device='cuda')
        n_main = len(c['moves'])
        for i, a in enumerate(acts):
            if i == n_main or (i == len(acts) - 1 and len(acts) == n_main and False):
                pass
            if i == n_main:
                for bad in c.get('then_raises', []):
                    if bad is None or gg.game_ended(s):
                        continue  # GoEnv-level "game over" assertion, covered in test_env
                    b = bad[0] * size + bad[1] if isinstance(bad, list) else bad
                    with pytest.raises(AssertionError):
                        gg.next_state(s, b)
            s2 = gg.next_state(s, int(a))
            assert np.array_equal(s2.cpu().numpy(), states[i]), (n, i)
            s = s2
        if len(acts) == n_main:
            for bad in c.get('then_raises', []):
                if bad is None or gg.game_ended(s):
                    continue
                b = bad[0] * size + bad[1] if isinstance(bad, list) else bad
                with pytest.raises(AssertionError):
                    gg.next_state(s, b)


def test_random_games_golden(gg, golden):
    """Seeded full games recorded from the reference: hashes of every ply, sampled states, areas, canonical."""
    z = golden('random_games')
    games = sorted({k.split('/')[0] for k in z.files})
    for gname in games:
        size = int(gname.split('_')[0][1:])
        acts, hashes = z[gname + '/actions'], z[gname + '/hashes']
        samp = {int(p): i for i, p in enumerate(z[gname + '/sample_ply'])}
        s = torch.zeros((1, 6, size, size), dtype=torch.uint8, device='cuda')
        for ply, a in enumerate(acts):
            s = gg.batch_next_states(s, torch.tensor([int(a)], device='cuda', dtype=torch.int32))
            sn = s[0].cpu().numpy()
            assert h64(sn) == hashes[ply], (gname, ply)
            if ply in samp:
                i = samp[ply]
                assert np.array_equal(sn, z[gname + '/sample_states'][i])
                b, w = gg.batch_areas(s)
                assert [int(b[0]), int(w[0])] == list(z[gname + '/sample_areas'][i]), (gname, ply)
                assert np.array_equal(gg.canonical_form(s[0]).cpu().numpy(), z[gname + '/sample_canonical'][i])
                assert np.array_equal(gg.invalid_moves(s[0]).cpu().numpy(), z[gname + '/sample_invalid_moves'][i])
        assert float(gg.winning(s[0], 0)) == float(z[gname + '/winning_komi0'])
        assert float(gg.winning(s[0], 6.5)) == float(z[gname + '/winning_komi6p5'])


def test_children_golden(gg, golden):
    z = golden('children')
    for key in sorted({k.split('/')[0] for k in z.files}):
        st = dev(z[key + '/state'])
        for canon, name in ((False, '/children'), (True, '/children_canonical')):
            got = gg.children(st, canonical=canon, padded=True).cpu().numpy()
            assert np.array_equal(got, z[key + name]), (key, canon)


def test_batch_with_passes_golden(gg, golden):
    """Mixed pass / move batches: stacked next_state semantics for every game."""
    z = golden('batch_passes')
    for size in (5, 9, 19):
        st, acts = dev(z['n%d/states' % size]), dev(z['n%d/actions' % size])
        for canon, name in ((False, 'next'), (True, 'next_canonical')):
            got = gg.batch_next_states(st, acts, canonical=canon).cpu().numpy()
            assert np.array_equal(got, z['n%d/%s' % (size, name)]), (size, canon)


def test_rollout_golden(gg, golden):
    """Device sampler + fused rollout reproduce the reference-replayed trajectories of the build's RNG."""
    z = golden('rollout')
    for size in (5, 9, 19):
        k = 'n%d/' % size
        rng0 = z[k + 'rng0']
        B, plies = len(rng0), int(z[k + 'plies'])
        rng = gg.rng_seed(B, int(z[k + 'seed']))
        assert np.array_equal(rng.cpu().numpy().view(np.uint64), rng0)
        st = torch.zeros((B, 6, size, size), dtype=torch.uint8, device='cuda')
        last = torch.empty(B, dtype=torch.int32, device='cuda')
        steps = torch.zeros(B, dtype=torch.int64, device='cuda')
        gg.batch_rollout(st, rng, plies, True, last, steps)
        assert np.array_equal(st.cpu().numpy(), z[k + 'final_states']), size
        assert np.array_equal(rng.cpu().numpy().view(np.uint64), z[k + 'rng_final'])
        assert np.array_equal(last.cpu().numpy(), z[k + 'last_actions'])
        assert int(steps.min()) == plies and int(steps.max()) == plies


@pytest.mark.parametrize('size', list(range(2, 20)))
def test_rollout_vs_oracle_all_sizes(gg, oracle, size):
    """Fused rollout vs the C oracle with the same generator, every board size the ABI accepts."""
    B, plies = 96, 3 * size * size + 10
    rng_np = oracle.rng_seed(1234 + size, B)
    rng = gg.rng_seed(B, 1234 + size)
    assert np.array_equal(rng.cpu().numpy().view(np.uint64), rng_np)
    st = torch.zeros((B, 6, size, size), dtype=torch.uint8, device='cuda')
    want = np.zeros((B, 6, size, size), dtype=np.uint8)
    # in chunks so that states are compared at many depths, also through game ends / auto-resets
    for chunk in (1, 2, 5, plies // 3, plies // 3, plies - 8 - 2 * (plies // 3)):
        gg.batch_rollout(st, rng, chunk, True)
        want, rng_np, _ = oracle.batch_rollout(want, rng_np, chunk, True)
        got = st.cpu().numpy()
        bad = np.nonzero((got != want).reshape(B, -1).any(axis=1))[0]
        assert len(bad) == 0, (size, chunk, bad[:5])
        assert np.array_equal(rng.cpu().numpy().view(np.uint64), rng_np)


@pytest.mark.parametrize('size,B,plies', [(9, 4096, 150), (19, 2048, 500)])
def test_step_api_vs_oracle_large(gg, oracle, size, B, plies):
    """gg_batch_sample_actions + gg_batch_next_states (out-of-place API) vs oracle replay of the same actions,
    plus areas / invalid-mask / children spot checks on the way."""
    rng = gg.rng_seed(B, 99)
    st = torch.zeros((B, 6, size, size), dtype=torch.uint8, device='cuda')
    want = np.zeros((B, 6, size, size), dtype=np.uint8)
    for ply in range(plies):
        ended = gg.batch_game_ended(st).bool()
        if bool(ended.any()):
            st[ended] = 0
            want[ended.cpu().numpy()] = 0
        acts = gg.batch_sample_actions(st, rng)
        nxt, status = gg.batch_next_states(st, acts, canonical=False, check=False)
        assert int(status.abs().sum()) == 0
        if ply % 25 == 0 or ply == plies - 1:
            w2, st_or = oracle.batch_next_states(want, acts.cpu().numpy())
            assert int(np.abs(st_or).sum()) == 0
            assert np.array_equal(nxt.cpu().numpy(), w2), (size, ply)
            want = w2
            b, w = gg.batch_areas(nxt)
            ob, ow = oracle.batch_areas(want)
            assert np.array_equal(b.cpu().numpy(), ob) and np.array_equal(w.cpu().numpy(), ow)
            from gymgo_amd import state_utils
            m = state_utils.batch_compute_invalid_moves(nxt[:64], None, None).cpu().numpy()
            for i in range(0, 64, 9):
                player = 1 - int(want[i, 2, 0, 0])
                assert np.array_equal(m[i], oracle.compute_invalid_moves(want[i], player)), (size, ply, i)
        else:
            want = nxt.cpu().numpy()
        st = nxt
    kids = gg.batch_children(st[:8], canonical=True).cpu().numpy()
    assert np.array_equal(kids, oracle.batch_children(want[:8], True))
    kids = gg.batch_children(st[8:12], canonical=False).cpu().numpy()
    assert np.array_equal(kids, oracle.batch_children(want[8:12], False))


@pytest.mark.parametrize('N,B,plies', [(19, 64, 90), (19, 8200, 100), (13, 4097, 80), (9, 300, 120), (5, 33, 50), (2, 17, 16)])
@pytest.mark.parametrize('canonical', [False, True])
def test_next_states_with_workspace_matches_oracle(gg, oracle, N, B, plies, canonical):
    """gg_batch_next_states_ws: a rollout through the out-of-place step API that feeds every output back as the next
    input, with the caller-owned workspace (liberty classes of the last outputs, reused when a board's stones match
    exactly).  Every call against the oracle - states and status - while the test also corrupts moves (refused rows
    pass through), swaps boards behind the workspace's back (those must be analysed afresh) and restarts finished
    games; both canonical settings; then the workspace really is the tracked form of the last output."""
    gen = np.random.default_rng(5 + N)
    st = torch.zeros((B, 6, N, N), dtype=torch.uint8, device='cuda')
    rng = gg.rng_seed(B, 5 + N)
    ws = gg.next_states_workspace(B, N)
    out = torch.empty_like(st)
    status = torch.empty(B, dtype=torch.int32, device='cuda')
    want = np.zeros((B, 6, N, N), np.uint8)
    hits = 0
    for t in range(plies):
        a = gg.batch_sample_actions(st, rng).cpu().numpy().copy()
        if t % 7 == 3:
            idx = gen.choice(B, max(1, B // 10), replace=False)
            a[idx] = gen.integers(-2, N * N + 3, size=len(idx))
        if t % 11 == 5:
            idx = gen.choice(B, max(1, B // 8), replace=False)
            src = gen.choice(B, len(idx))
            st[torch.as_tensor(idx, device='cuda')] = st[torch.as_tensor(src, device='cuda')]
            want[idx] = want[src]
        if t % 9 == 8:   # how many boards will take their classes from the workspace in this call
            hits += int((gg.batch_pack(st)[:, :2 * N] == ws[:, :2 * N]).all(dim=1).sum())
        gg.batch_next_states(st, torch.from_numpy(a).cuda(), canonical=canonical, check=False, out=out, status=status, workspace=ws)
        want, wst = oracle.batch_next_states_mt(want, a, canonical)
        assert np.array_equal(status.cpu().numpy(), wst), (N, t)
        assert np.array_equal(out.cpu().numpy(), want), (N, t, canonical)
        st, out = out, st
        if t == plies - 1:   # the workspace is the tracked form of what the call just wrote (refused rows: untouched)
            moved = (status == 0)
            assert torch.equal(ws[moved], gg.batch_track(st)[moved])
        over = want[:, 5, 0, 0] == 1
        if over.any() and t % 5 == 0:
            want[over] = 0
            st[torch.from_numpy(over).cuda()] = 0
    assert hits > B * (plies // 9) // 2                     # the workspace was used, not just carried along
    with pytest.raises(ValueError):
        gg.batch_next_states(st, torch.zeros(B, dtype=torch.int32, device='cuda'), check=False, out=out, workspace=ws[:, :-1])


def test_illegal_moves_status_and_passthrough(gg, oracle):
    size, B = 9, 256
    rng = gg.rng_seed(B, 5)
    st = torch.zeros((B, 6, size, size), dtype=torch.uint8, device='cuda')
    gg.batch_rollout(st, rng, 40, True)
    s_np = st.cpu().numpy()
    acts = np.zeros(B, dtype=np.int32)
    for b in range(B):
        occ = np.flatnonzero(s_np[b, 3].ravel() == 1)
        acts[b] = occ[b % len(occ)] if (b % 2 == 0 and len(occ)) else size * size
    acts[3], acts[5] = -1, size * size + 1   # out of range
    out, status = gg.batch_next_states(st, dev(acts), check=False)
    want, wst = oracle.batch_next_states(s_np, acts)
    assert np.array_equal(status.cpu().numpy(), wst)
    assert np.array_equal(out.cpu().numpy(), want)
    assert int(status.sum()) > 50
    with pytest.raises(AssertionError):
        gg.batch_next_states(st, dev(acts))


@pytest.mark.parametrize('size', [9, 13, 19, 7])
def test_next_states_and_env_step_on_both_sides_of_the_small_batch_take_over(gg, oracle, size):
    """Up to 16 / 8 pairs per CU (9x9 / larger) gg_batch_next_states runs the straight one-pair-per-wave kernel and
    gg_batch_env_step goes out as four-wave workgroups; above, the pipelined / single-wave forms: mid-game boards with a mix
    of legal points, occupied points, passes and out-of-range moves (canonical on and off) at the batch sizes around the
    take-over and at ragged workgroups - states and status against the oracle; one env step against the oracle's ply."""
    from gymgo_amd import _lib
    cus = int(_lib.lib().gg_device_cus())
    edge = 2 * cus * (16 if size <= 9 else 8)
    for B in (1, 2, 3, 5, 7, 8, 9, 15, 17, edge - 1, edge, edge + 1, edge + 2, edge + 9):
        rng = gg.rng_seed(B, 17 + B)
        st = torch.zeros((B, 6, size, size), dtype=torch.uint8, device='cuda')
        gg.batch_rollout(st, rng, 3 * size, False)
        s_np = st.cpu().numpy()
        acts = gg.batch_sample_actions(st, rng).cpu().numpy().astype(np.int32)
        r = np.random.RandomState(B)
        for b in range(0, B, 3):      # every third game: an occupied / invalid point where there is one
            occ = np.flatnonzero(s_np[b, 3].ravel() == 1)
            if len(occ):
                acts[b] = occ[r.randint(len(occ))]
        acts[::7] = size * size       # passes
        if B > 4:
            acts[1], acts[4] = -1, size * size + 1
        for canonical in (False, True):
            out, status = gg.batch_next_states(st, dev(acts), canonical=canonical, check=False)
            want, wst = oracle.batch_next_states_mt(s_np, acts, canonical)
            assert np.array_equal(status.cpu().numpy(), wst), (size, B, canonical)
            assert np.array_equal(out.cpu().numpy(), want), (size, B, canonical)
        es, er = st.clone(), rng.clone()
        want2, rng2, last2 = oracle.batch_rollout_mt(s_np, rng.cpu().numpy().view(np.uint64).copy(), 1, True)
        rewards, dones, status, taken = gg.batch_env_step(es, None, er, 0.5, 'real', True)
        assert np.array_equal(es.cpu().numpy(), want2) and np.array_equal(taken.cpu().numpy(), last2), (size, B)
        assert np.array_equal(er.cpu().numpy().view(np.uint64), rng2) and int(status.abs().sum()) == 0


def test_unaligned_views(gg, oracle):
    """Boards that start at odd byte offsets / non-16-byte-aligned bases (head and tail byte paths)."""
    size, B = 19, 33
    rng = gg.rng_seed(B, 17)
    base = torch.zeros(B * 6 * size * size + 64, dtype=torch.uint8, device='cuda')
    for off in (0, 1, 7, 15, 16, 33):
        st = base[off:off + B * 6 * size * size].view(B, 6, size, size)
        st.zero_()
        gg.batch_rollout(st, rng, 120, True)
        s_np = st.cpu().numpy()
        acts = gg.batch_sample_actions(st, rng)
        for off2 in (3, 16):
            outbuf = torch.full((B * 6 * size * size + 64,), 7, dtype=torch.uint8, device='cuda')
            out = outbuf[off2:off2 + B * 6 * size * size].view(B, 6, size, size)
            status = torch.empty(B, dtype=torch.int32, device='cuda')
            from gymgo_amd import _lib
            code = _lib.lib().gg_batch_next_states(st.data_ptr(), acts.data_ptr(), out.data_ptr(), status.data_ptr(),
                                                   B, size, 0, None)
            assert code == 0
            want, _ = oracle.batch_next_states(s_np, acts.cpu().numpy())
            assert np.array_equal(out.cpu().numpy(), want), (off, off2)
            ob = outbuf.cpu().numpy()
            assert (ob[:off2] == 7).all() and (ob[off2 + B * 6 * size * size:] == 7).all(), 'wrote outside the batch'


def test_empty_batch_and_bad_args(gg):
    from gymgo_amd import _lib
    L = _lib.lib()
    assert L.gg_version() == _lib.ABI_VERSION == 5
    assert L.gg_batch_next_states(None, None, None, None, 0, 9, 0, None) == 0
    assert L.gg_batch_next_states(None, None, None, None, 4, 9, 0, None) == -2
    assert L.gg_batch_next_states(None, None, None, None, 4, 20, 0, None) == -1
    assert L.gg_batch_areas(None, None, None, 4, 1, None) == -1
    assert L.gg_device_cus() > 0


@pytest.mark.parametrize('N,B', [(19, 12288), (19, 8201), (13, 10240), (11, 9000), (9, 12345), (6, 8192)])
def test_multi_ply_kernel_at_its_dispatch_sizes(gg, oracle, N, B):
    """The multi-ply kernel (liberty classes carried across the plies of a launch) serves gg_batch_rollout /
    _packed / gg_batch_play_moves from 8 192 games and two plies per launch up, with 8, 10 or 12 boards per wave
    depending on the batch.  Here at those sizes, every row capacity (N == capacity and N < capacity), odd batches:
    rollouts from the empty board with and without auto-reset in launches of 8 / 9 / 33 / 64 plies, ORACLE replay of
    every 32nd game (states, generator, last action); the packed form must equal the byte-plane form for every game;
    then a recorded continuation, a third of the games corrupted, replayed in one launch (given moves) vs the oracle."""
    idx = np.arange(0, B, 32)
    idx_t = torch.as_tensor(idx, device='cuda')
    for auto in (True, False):
        rng = gg.rng_seed(B, 7 + N)
        rng_np = np.array([oracle.lib().gg_oracle_rng_seed(7 + N, int(i)) for i in idx], dtype=np.uint64)
        st = torch.zeros((B, 6, N, N), dtype=torch.uint8, device='cuda')
        want = np.zeros((len(idx), 6, N, N), np.uint8)
        pk = gg.batch_pack(st)
        prng = rng.clone()
        la = torch.empty(B, dtype=torch.int32, device='cuda')
        sd = torch.zeros(B, dtype=torch.int64, device='cuda')
        total = 0
        for plies in (8, 9, 33, 64 if N > 6 else 20):
            gg.batch_rollout(st, rng, plies, auto, la, sd)
            gg.batch_rollout_packed(pk, prng, plies, auto)
            want, rng_np, wl = oracle.batch_rollout_mt(want, rng_np, plies, auto)
            total += plies
            assert np.array_equal(st[idx_t].cpu().numpy(), want), ('rollout', N, auto, total)
            assert np.array_equal(rng[idx_t].cpu().numpy().view(np.uint64), rng_np), ('rng', N, auto, total)
            assert np.array_equal(la[idx_t].cpu().numpy(), wl), ('last', N, auto, total)
            assert torch.equal(gg.batch_unpack(pk, N), st) and torch.equal(prng, rng), ('packed', N, auto, total)
        if auto:
            assert int(sd.min()) == total
    # given moves: record a continuation ply by ply (per-ply kernel), corrupt a third of the games, replay in one launch
    T = 24
    rec = torch.empty((B, T), dtype=torch.int32, device='cuda')
    start, tmp, r2 = st.clone(), st.clone(), gg.rng_seed(B, 99)
    for t in range(T):
        gg.batch_rollout(tmp, r2, 1, False, la, None)
        rec[:, t] = la
    moves = rec.cpu().numpy().copy()
    gen = np.random.default_rng(N)
    for i in gen.choice(B, max(1, B // 3), replace=False):
        moves[i, gen.integers(0, T)] = gen.integers(-2, N * N + 2)
    host = start[idx_t].cpu().numpy()
    exp, played = host.copy(), np.zeros(len(idx), np.int32)
    for j, i in enumerate(idx):
        s = host[j]
        for t in range(T):
            a = int(moves[i, t])
            if s[5, 0, 0] or a < 0 or a > N * N or (a < N * N and s[3].reshape(-1)[a]):
                break
            s = oracle.next_state(s, a)
            played[j] += 1
        exp[j] = s
    pk = gg.batch_pack(start)
    got = gg.batch_play_moves(start, torch.from_numpy(moves).cuda())
    assert np.array_equal(got[idx_t].cpu().numpy(), played) and np.array_equal(start[idx_t].cpu().numpy(), exp), ('play_moves', N)
    got2 = gg.batch_play_moves(pk, torch.from_numpy(moves).cuda())
    assert torch.equal(got2, got) and torch.equal(gg.batch_unpack(pk, N), start)
    untouched = torch.from_numpy((moves == rec.cpu().numpy()).all(axis=1)).cuda()
    assert torch.equal(start[untouched], tmp[untouched])     # uncorrupted games: the replay equals the recording run


@pytest.mark.parametrize('size', list(range(2, 20)))
def test_areas_vs_oracle_all_sizes(gg, oracle, size):
    """gg_batch_areas (sixteen boards per wave, planes 0 / 1 staged eight boards at a time): boards from every game
    phase incl. finished games and empty boards, a batch that is no multiple of 16, read through an UNALIGNED view
    (board stride 6 N^2 bytes, first board 3 boards in) - every game vs the oracle's areas (gym_go/gogame.py:275-310)."""
    N, B = size, 1000 + size
    st = gg.batch_init_state(B, N, device='cuda')
    rng = gg.rng_seed(B, 40 + N)
    for g in range(4):
        hi = (g + 1) * 250 + (N if g == 3 else 0)
        gg.batch_rollout(st[g * 250:hi], rng[g * 250:hi], (g * N * N) // 3, False)
    view = st[3:]
    black, white = gg.batch_areas(view)
    ob, ow = oracle.batch_areas_mt(view.cpu().numpy())
    assert np.array_equal(black.cpu().numpy(), ob) and np.array_equal(white.cpu().numpy(), ow)
    assert int(black[:200].sum()) == 0 and int(white[:200].sum()) == 0            # empty boards: nobody's area
    one = gg.batch_areas(view[5:6])                                               # a single board: a wave of one
    assert int(one[0][0]) == int(ob[5]) and int(one[1][0]) == int(ow[5])


def test_update_pieces_standalone(gg, oracle):
    """state_utils.update_pieces / batch_update_pieces (gg_batch_update_pieces): stones after capture resolution equal
    planes 0/1 of the oracle's next_state; killed groups are reported per group in raster order."""
    from gymgo_amd import state_utils
    size, B = 9, 400
    rng = gg.rng_seed(B, 123)
    st = torch.zeros((B, 6, size, size), dtype=torch.uint8, device='cuda')
    gg.batch_rollout(st, rng, 70, True)
    acts = gg.batch_sample_actions(st, rng).cpu().numpy()
    s_np = st.cpu().numpy()
    keep = np.flatnonzero((acts < size * size) & (s_np[:, 5, 0, 0] == 0))
    want, _ = oracle.batch_next_states(s_np[keep], acts[keep])
    placed = s_np[keep].copy()
    players = placed[:, 2, 0, 0].astype(np.int32)
    rc = np.stack([acts[keep] // size, acts[keep] % size], axis=1)
    placed[np.arange(len(keep)), players, rc[:, 0], rc[:, 1]] = 1
    adj = [state_utils.adj_data(placed[i], rc[i], int(players[i]))[0] for i in range(len(keep))]
    batch = placed.astype(np.float64)          # the reference's container: float64, mutated in place
    killed = state_utils.batch_update_pieces(np.arange(len(keep)), batch, adj, players)
    assert np.array_equal(batch[:, :2].astype(np.uint8), want[:, :2])
    n_capt = 0
    for i in range(len(keep)):
        removed = int(placed[i, 1 - players[i]].sum() - want[i, 1 - players[i]].sum())
        assert sum(len(g) for g in killed[i]) == removed
        n_capt += removed > 0
    assert n_capt > 10
    # single-state form on a device tensor, multi-group capture: the 4 black stones around (1,1) die one by one
    s = torch.zeros((6, 5, 5), dtype=torch.uint8, device='cuda')
    for r, c in ((0, 1), (1, 0), (1, 2), (2, 1)):
        s[0, r, c] = 1
    for r, c in ((0, 0), (0, 2), (2, 0), (2, 2), (1, 3), (3, 1), (0, 3), (3, 0)):
        s[1, r, c] = 1
    s[1, 1, 1] = 1   # white plays the centre: all four black stones have no liberty left
    groups = state_utils.update_pieces(s, np.array([[0, 1], [2, 1], [1, 0], [1, 2]]), 1)
    assert int(s[0].sum()) == 0 and len(groups) == 4 and all(len(g) == 1 for g in groups)
    assert [tuple(g[0]) for g in groups] == [(0, 1), (1, 0), (1, 2), (2, 1)]


@pytest.mark.parametrize('size', [2, 5, 9, 13, 19])
def test_packed_format_roundtrip(gg, size):
    """gg_batch_pack_states / gg_batch_unpack_states: row-mask words match a NumPy packing of the same states and
    unpack(pack(s)) == s bit-exactly (odd batch size, mid-game + finished + empty boards)."""
    B = 257
    rng = gg.rng_seed(B, 31 + size)
    st = torch.zeros((B, 6, size, size), dtype=torch.uint8, device='cuda')
    gg.batch_rollout(st[:200], rng[:200].clone(), 2 * size * size, False)      # some of these end (frozen, DONE set)
    gg.batch_rollout(st[200:250], rng[200:250].clone(), size * size // 2, True)
    packed = gg.batch_pack(st)
    assert packed.shape == (B, 3 * size + 1) and packed.dtype == torch.int32
    s = st.cpu().numpy()
    weights = (1 << np.arange(size)).astype(np.int64)
    want = np.concatenate([(s[:, p].astype(np.int64) * weights).sum(axis=2) for p in (0, 1, 3)] +
                          [(s[:, 2, 0, 0] | (s[:, 4, 0, 0] << 1) | (s[:, 5, 0, 0] << 2)).astype(np.int64)[:, None]], axis=1)
    assert np.array_equal(packed.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, want)
    back = gg.batch_unpack(packed, size)
    assert torch.equal(back, st)
    assert int(s[:, 5, 0, 0].sum()) > 0 or size > 9


def test_hipgraph_capture_of_per_ply_loop(gg):
    """The C-ABI launches go to torch's current stream, so a K-ply loop of 1-ply launches plus out-of-place steps can be
    captured into a hipGraph (torch.cuda.CUDAGraph) and replayed; the replay is bit-exact with the fused rollout."""
    B, N, K = 1024, 9, 12
    st = torch.zeros((B, 6, N, N), dtype=torch.uint8, device='cuda')
    rng = gg.rng_seed(B, 77)
    gg.batch_rollout(st, rng, 30, True)
    ref_st, ref_rng = st.clone(), rng.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gg.batch_rollout(st, rng, 1, True)
        acts = gg.batch_sample_actions(st, rng)
        nxt, status = gg.batch_next_states(st, acts, check=False)
    torch.cuda.current_stream().wait_stream(side)
    st.copy_(ref_st); rng.copy_(ref_rng)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(K):
            gg.batch_rollout(st, rng, 1, True)
        acts = gg.batch_sample_actions(st, rng)
        nxt, status = gg.batch_next_states(st, acts, check=False)
    for _ in range(2):   # replay twice from the same start: identical results
        st.copy_(ref_st); rng.copy_(ref_rng)
        graph.replay()
        torch.cuda.synchronize()
        want, wr = ref_st.clone(), ref_rng.clone()
        gg.batch_rollout(want, wr, K, True)
        assert torch.equal(st, want)
        wa = gg.batch_sample_actions(want, wr)
        assert torch.equal(acts, wa) and torch.equal(rng, wr)
        wn, ws = gg.batch_next_states(want, wa, check=False)
        assert torch.equal(nxt, wn) and int(status.sum()) == 0
