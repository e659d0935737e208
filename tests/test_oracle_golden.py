"""CPU: both oracles (C restatement, NumPy/SciPy port) against the golden vectors recorded from the
real reference (tests/golden/make_golden.py).  No GPU, no /root/reference needed."""
import hashlib

import numpy as np
import pytest

from oracle import c_oracle, np_oracle


def h64(a):
    return np.frombuffer(hashlib.blake2b(np.ascontiguousarray(a).tobytes(), digest_size=8).digest(), dtype=np.uint64)[0]


def test_scripted_sequences(golden, scripted_cases):
    z = golden('scripted')
    assert len(scripted_cases) == 33
    for c in scripted_cases:
        n, size = c['name'], c['size']
        acts, states = z[n + '/actions'], z[n + '/states']
        s = np.zeros((6, size, size), np.uint8)
        n_main = len(c['moves'])

        def refuses(state):
            for bad in c.get('then_raises', []):
                if bad is None or state[5].all():
                    continue   # env-level "game already over" assertion (tests/test_env_host.py)
                b = bad[0] * size + bad[1] if isinstance(bad, list) else bad
                with pytest.raises(AssertionError):
                    c_oracle.next_state(state, b)
                with pytest.raises(AssertionError):
                    np_oracle.next_state(state.astype(float), b)
        for i, a in enumerate(acts):
            if i == n_main:
                refuses(s)
            s = c_oracle.next_state(s, int(a))
            assert np.array_equal(s, states[i]), (n, i)
            if size <= 7 and i % 3 == 0:
                assert np.array_equal(np_oracle.next_state(states[i - 1].astype(float) if i else
                                                           np.zeros((6, size, size)), int(a)).astype(np.uint8), states[i])
        if len(acts) == n_main:
            refuses(s)
        # the values the reference's own unit tests assert
        final, pins = states[n_main - 1], c.get('pins', {})
        if 'invd_count' in pins:
            assert int(final[3].sum()) == pins['invd_count']
        for r_, c_, v in pins.get('invd_at', []):
            assert final[3, r_, c_] == v
        if 'black' in pins:
            assert int(final[0].sum()) == pins['black']
        if 'white' in pins:
            assert int(final[1].sum()) == pins['white']
        if 'nonzero_total' in pins:
            assert int(np.count_nonzero(final)) == pins['nonzero_total']
        if 'done' in pins:
            assert int(final[5].all()) == pins['done']


def test_random_games(golden):
    z = golden('random_games')
    games = sorted({k.split('/')[0] for k in z.files})
    assert len(games) == 21
    for gname in games:
        size = int(gname.split('_')[0][1:])
        acts, hashes = z[gname + '/actions'], z[gname + '/hashes']
        samp = {int(p): i for i, p in enumerate(z[gname + '/sample_ply'])}
        s = np.zeros((6, size, size), np.uint8)
        for ply, a in enumerate(acts):
            s = c_oracle.next_state(s, int(a))
            assert h64(s) == hashes[ply], (gname, ply)
            if ply in samp:
                i = samp[ply]
                assert np.array_equal(s, z[gname + '/sample_states'][i])
                b, w = c_oracle.batch_areas(s[None])
                assert [int(b[0]), int(w[0])] == list(z[gname + '/sample_areas'][i])
                assert np.array_equal(c_oracle.canonical_form(s), z[gname + '/sample_canonical'][i])
                if size <= 9:
                    nb, nw = np_oracle.areas(s.astype(float))
                    assert [int(nb), int(nw)] == list(z[gname + '/sample_areas'][i])
        b, w = c_oracle.batch_areas(s[None])
        assert np.sign(float(b[0]) - float(w[0])) == float(z[gname + '/winning_komi0'])
        assert np.sign(float(b[0]) - float(w[0]) - 6.5) == float(z[gname + '/winning_komi6p5'])


def test_np_port_on_19x19_game(golden):
    """The cpu_baseline port replays one full recorded 19x19 game bit-exactly."""
    z = golden('random_games')
    acts, hashes = z['n19_g0/actions'][:160], z['n19_g0/hashes']
    s = np.zeros((6, 19, 19))
    for ply, a in enumerate(acts):
        s = np_oracle.next_state(s, int(a))
        assert h64(s.astype(np.uint8)) == hashes[ply], ply


def test_children(golden):
    z = golden('children')
    for key in sorted({k.split('/')[0] for k in z.files}):
        st = z[key + '/state']
        assert np.array_equal(c_oracle.batch_children(st[None], False)[0], z[key + '/children']), key
        assert np.array_equal(c_oracle.batch_children(st[None], True)[0], z[key + '/children_canonical']), key


def test_batch_with_passes(golden):
    z = golden('batch_passes')
    for size in (5, 9, 19):
        st, acts = z['n%d/states' % size], z['n%d/actions' % size]
        assert (acts == size * size).sum() >= 4
        for canon, name in ((False, 'next'), (True, 'next_canonical')):
            got, status = c_oracle.batch_next_states(st, acts, canon)
            assert not status.any()
            assert np.array_equal(got, z['n%d/%s' % (size, name)]), (size, canon)


def test_rollout_sampler(golden):
    z = golden('rollout')
    for size in (5, 9, 19):
        k = 'n%d/' % size
        rng0 = z[k + 'rng0']
        assert np.array_equal(c_oracle.rng_seed(int(z[k + 'seed']), len(rng0)), rng0)
        st, rng, last = c_oracle.batch_rollout(np.zeros((len(rng0), 6, size, size), np.uint8), rng0,
                                               int(z[k + 'plies']), True)
        assert np.array_equal(st, z[k + 'final_states'])
        assert np.array_equal(rng, z[k + 'rng_final'])
        assert np.array_equal(last, z[k + 'last_actions'])


def test_frozen_when_no_auto_reset():
    st = np.zeros((2, 6, 5, 5), np.uint8)
    rng = c_oracle.rng_seed(3, 2)
    st, rng, _ = c_oracle.batch_rollout(st, rng, 400, False)
    assert st[:, 5].all()      # both games ended and stayed ended
    st2, rng2, last = c_oracle.batch_rollout(st, rng, 5, False)
    assert np.array_equal(st2, st) and np.array_equal(rng2, rng) and (last == -1).all()


def test_update_pieces_arbitrary_positions(golden):
    """state_utils.update_pieces (gym_go/state_utils.py:159-180) as recorded from the reference on positions that are
    not reachable by legal play (corners, duplicates, long location lists): the C restatement's stand-alone entry."""
    z = golden('extras')
    n = int(z['up/count'])
    assert n >= 150
    for i in range(n):
        k = 'up/%d/' % i
        state, adj, player = z[k + 'state'], z[k + 'adj'], int(z[k + 'player'])
        after, killed, groups = c_oracle.update_pieces(state, adj[:, 0] * state.shape[-1] + adj[:, 1], player)
        assert np.array_equal(after, z[k + 'after']) and np.array_equal(killed, z[k + 'killed']), i
        assert groups == int(z[k + 'groups']), i


@pytest.mark.parametrize('size,depths', [(3, (2, 5, 9)), (5, (4, 12, 22, 30)), (9, (10, 35, 60, 85)), (13, (30, 90, 150)),
                                         (19, (60, 180, 300))])
def test_c_restatement_vs_scipy_port_on_every_child(size, depths):
    """Differential pinning beyond the recorded goldens: the C restatement (the oracle the GPU tests use) against the
    NumPy / SciPy port (same scipy.ndimage call structure as gym_go/state_utils.py, itself pinned to the reference by the
    goldens above) on EVERY action of positions from every game phase - each legal move's successor incl. captures, ko and
    the suicide boundary, canonical form on and off, and the areas of every successor."""
    for depth in depths:
        rng = c_oracle.rng_seed(1000 + depth, 4)
        states, rng, _ = c_oracle.batch_rollout(np.zeros((4, 6, size, size), np.uint8), rng, depth, False)
        for s in states:
            if s[5, 0, 0]:
                continue                                   # children of a finished game: undefined in the reference
            for canon in (False, True):
                kids = c_oracle.batch_children(s[None], canon)[0]
                for a in range(size * size + 1):
                    if a < size * size and s[3].reshape(-1)[a]:
                        assert not kids[a].any()
                        continue
                    want = np_oracle.next_state(s.astype(float), a, canon)
                    assert np.array_equal(kids[a], want.astype(np.uint8)), (size, depth, a, canon)
                    if not canon and a % 7 == 0:
                        b, w = c_oracle.batch_areas(kids[a][None])
                        nb, nw = np_oracle.areas(want)
                        assert (int(b[0]), int(w[0])) == (int(nb), int(nw)), (size, depth, a)


def test_weighted_sampler_oracle_distribution_matches_the_reference(golden):
    """CPU: the exact integer sampler of oracle/gg_oracle.c (gg_oracle_sample_weighted) draws from the distribution the
    REFERENCE draws from - its own sklearn l1-normalisation of the weights (gym_go/gogame.py:391), recorded per weight
    vector in tests/golden/policy.npz together with 20 000 of the reference's own seeded draws; and the committed
    (states, weights, generator) -> actions vectors reproduce."""
    from oracle import c_oracle
    z = golden('policy')
    n = 20000
    for c in range(int(z['case/count'])):
        k = 'case/%d/' % c
        N = int(z[k + 'size'])
        state, w, p, href = z[k + 'state'], z[k + 'weights'], z[k + 'p_reference'], z[k + 'hist_reference']
        assert abs(p.sum() - 1) < 1e-9 and not p[np.append(state[3].ravel(), 0) == 1].any()
        acts, _ = c_oracle.batch_sample_weighted(np.broadcast_to(state, (n,) + state.shape), np.broadcast_to(w, (n, len(w))),
                                                 c_oracle.rng_seed(500 + c, n))
        hist = np.bincount(acts, minlength=N * N + 1)
        sigma = np.sqrt(n * p * (1 - p))
        assert np.all(np.abs(hist - n * p) <= 5 * sigma + 1.5), (c, str(z[k + 'kind']))
        assert not hist[p == 0].any()
        both = (hist + href) > 0
        chi2 = float((((hist - href) ** 2)[both] / (hist + href)[both]).sum())
        dof = int(both.sum()) - 1
        assert chi2 < dof + 6 * np.sqrt(2 * max(dof, 1)) + 10, (c, chi2, dof)
    for N in (5, 9, 19):
        k = 'exact/%d/' % N
        acts, rng1 = c_oracle.batch_sample_weighted(z[k + 'states'], z[k + 'weights'], z[k + 'rng0'])
        assert np.array_equal(acts, z[k + 'actions']) and np.array_equal(rng1, z[k + 'rng1'])
        # the quantisation is monotone and scale-free: multiplying a row by a power of two never changes the draw
        acts2, _ = c_oracle.batch_sample_weighted(z[k + 'states'], z[k + 'weights'] * np.float32(2.0 ** 20), z[k + 'rng0'])
        big = np.abs(z[k + 'weights']).max(axis=1) > 1e-20      # (rows scaled by 1e-30 leave the normal range when un-scaled)
        assert np.array_equal(acts2[big], z[k + 'actions'][big])
