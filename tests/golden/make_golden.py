"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Run only in the build container (needs /root/reference):   python tests/golden/make_golden.py
Outputs (committed; data only - inputs and expected outputs, no reference source):
  scripted.npz   per-step uint8 states / rewards / done / info for every case of scripted_cases.json,
                 driven through the reference's own GoEnv (gym.make('gym_go:go-v0', ...)).
  random_games.npz   seeded uniform-random full games (3..19): actions, 64-bit state hashes per ply,
                 sampled full states with areas, canonical forms and ko-sensitive invalid vectors.
  children.npz   gogame.children(state, canonical in {F,T}, padded=True) at several plies.
  batch_passes.npz   mixed pass / move batches answered by STACKED gogame.next_state.
  rollout.npz    actions drawn by the build's counter-based sampler (oracle/gg_oracle.c), replayed
                 through the reference's next_state: final states after K plies.
  extras.npz     the helpers nothing else pins: gogame.str (text of several positions), gogame.all_symmetries (the 8
                 views of a [C,N,N] image), state_utils.update_pieces on ARBITRARY (not legally reachable) positions
                 incl. every corner / edge, with the location lists the reference's callers pass and longer ones.

  policy.npz     gogame.random_weighted_action / random_action (gym_go/gogame.py:385-404): for several weight vectors (with
                 and without an invalid-move mask) the probability vector the REFERENCE draws from (its own
                 sklearn l1-normalisation) and the histogram of 20 000 seeded draws of the reference itself; plus
                 mid-game states whose plane 3 masks the weights, and the actions the build's exact sampler
                 (oracle/gg_oracle.c, gg_oracle_sample_weighted) draws for them (these pin kernel == oracle on the
                 GPU box; the reference pins the DISTRIBUTION).

  python tests/golden/make_golden.py [name ...]     (no name = all of them)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle', 'ref_harness'))

import refimport  # noqa: E402
from oracle import c_oracle  # noqa: E402


def u8(x):
    return np.asarray(x).astype(np.uint8)


def h64(state_u8):
    return np.frombuffer(hashlib.blake2b(np.ascontiguousarray(state_u8).tobytes(), digest_size=8).digest(),
                         dtype=np.uint64)[0]


def as_action(m, size):
    if m is None:
        return size * size
    if isinstance(m, (list, tuple)):
        return m[0] * size + m[1]
    return int(m)


def scripted(gym, gogame):
    cases = json.load(open(os.path.join(HERE, 'scripted_cases.json')))['cases']
    out = {}
    for c in cases:
        size = c['size']
        env = gym.make('gym_go:go-v0', size=size, komi=c.get('komi', 0), reward_method=c.get('reward_method', 'real'))
        env.reset()
        states, rewards, dones, turns, invds, passed, libs = [], [], [], [], [], [], []
        moves = list(c['moves'])
        n_main = len(moves)
        moves += c.get('continue', [])
        raises_ok = []
        for i, m in enumerate(moves):
            if i == n_main:
                for bad in c.get('then_raises', []):
                    try:
                        env.step(tuple(bad) if isinstance(bad, list) else bad)
                        raises_ok.append(0)
                    except Exception:
                        raises_ok.append(1)
            s, r, d, info = env.step(tuple(m) if isinstance(m, list) else m)
            states.append(u8(s)); rewards.append(float(r)); dones.append(int(d))
            turns.append(int(info['turn'])); invds.append(u8(info['invalid_moves']))
            passed.append(int(info['prev_player_passed']))
            libs.append(gogame.num_liberties(s))
        if len(moves) == n_main:
            for bad in c.get('then_raises', []):
                try:
                    env.step(tuple(bad) if isinstance(bad, list) else bad)
                    raises_ok.append(0)
                except Exception:
                    raises_ok.append(1)
        assert all(raises_ok), c['name']
        # the reference's own pins hold on the reference (sanity of the lifted scripts)
        final = states[n_main - 1]
        pins = c.get('pins', {})
        if 'invd_count' in pins:
            assert int(final[3].sum()) == pins['invd_count'], c['name']
        for r_, c_, v in pins.get('invd_at', []):
            assert final[3, r_, c_] == v, c['name']
        if 'black' in pins:
            assert int(final[0].sum()) == pins['black'], c['name']
        if 'white' in pins:
            assert int(final[1].sum()) == pins['white'], c['name']
        if 'nonzero_total' in pins:
            assert int(np.count_nonzero(final)) == pins['nonzero_total'], c['name']
        if 'reward' in pins:
            assert rewards[n_main - 1] == pins['reward'], c['name']
        if 'done' in pins:
            assert dones[n_main - 1] == pins['done'], c['name']
        if 'rewards' in c:
            assert rewards[:n_main] == [float(x) for x in c['rewards']], c['name']
        if 'num_liberties' in c:
            assert [list(map(int, x)) for x in libs[:n_main]] == c['num_liberties'], c['name']
        n = c['name']
        out[n + '/actions'] = np.array([as_action(m, size) for m in moves], dtype=np.int32)
        out[n + '/states'] = np.stack(states)
        out[n + '/rewards'] = np.array(rewards)
        out[n + '/dones'] = np.array(dones, dtype=np.int32)
        out[n + '/turns'] = np.array(turns, dtype=np.int32)
        out[n + '/invalid_moves'] = np.stack(invds)
        out[n + '/prev_passed'] = np.array(passed, dtype=np.int32)
        out[n + '/num_liberties'] = np.array(libs, dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, 'scripted.npz'), **out)
    print('scripted: %d cases' % len(cases))


def random_games(gogame, govars):
    rng = np.random.default_rng(7)
    out = {}
    plan = {3: 4, 5: 4, 7: 4, 9: 4, 13: 2, 19: 3}
    for size, games in plan.items():
        for g in range(games):
            s = gogame.init_state(size)
            actions, hashes, samp_idx, samp_states, samp_areas, samp_canon, samp_invalid = [], [], [], [], [], [], []
            for ply in range(6 * size * size):
                if gogame.game_ended(s):
                    break
                valid = np.flatnonzero(np.append(s[govars.INVD_CHNL].ravel(), 0) == 0)
                a = int(rng.choice(valid))
                s = gogame.next_state(s, a)
                actions.append(a)
                hashes.append(h64(u8(s)))
                if ply % (16 if size > 9 else 5) == 0 or gogame.game_ended(s):
                    samp_idx.append(ply)
                    samp_states.append(u8(s))
                    samp_areas.append([int(x) for x in gogame.areas(s)])
                    samp_canon.append(u8(gogame.canonical_form(s)))
                    samp_invalid.append(u8(gogame.invalid_moves(s)))
            k = 'n%d_g%d/' % (size, g)
            out[k + 'actions'] = np.array(actions, dtype=np.int32)
            out[k + 'hashes'] = np.array(hashes, dtype=np.uint64)
            out[k + 'sample_ply'] = np.array(samp_idx, dtype=np.int32)
            out[k + 'sample_states'] = np.stack(samp_states)
            out[k + 'sample_areas'] = np.array(samp_areas, dtype=np.int32)
            out[k + 'sample_canonical'] = np.stack(samp_canon)
            out[k + 'sample_invalid_moves'] = np.stack(samp_invalid)
            out[k + 'winning_komi0'] = np.array(gogame.winning(s, 0))
            out[k + 'winning_komi6p5'] = np.array(gogame.winning(s, 6.5))
    np.savez_compressed(os.path.join(HERE, 'random_games.npz'), **out)
    print('random_games: %d arrays' % len(out))


def children(gogame, govars):
    rng = np.random.default_rng(11)
    out = {}
    for size, plies_list in {3: (0, 3, 6), 5: (0, 9, 20), 7: (5, 30), 9: (12, 60), 19: (150,)}.items():
        s = gogame.init_state(size)
        done = 0
        for target in plies_list:
            while done < target and not gogame.game_ended(s):
                valid = np.flatnonzero(np.append(s[govars.INVD_CHNL].ravel(), 0) == 0)[:-1]
                if len(valid) == 0:
                    break
                s = gogame.next_state(s, int(rng.choice(valid)))
                done += 1
            k = 'n%d_p%d/' % (size, target)
            out[k + 'state'] = u8(s)
            out[k + 'children'] = u8(gogame.children(s, False, True))
            out[k + 'children_canonical'] = u8(gogame.children(s, True, True))
    np.savez_compressed(os.path.join(HERE, 'children.npz'), **out)
    print('children: %d arrays' % len(out))


def batch_passes(gogame, govars):
    rng = np.random.default_rng(13)
    out = {}
    for size, B in ((5, 24), (9, 32), (19, 12)):
        states, actions = [], []
        for b in range(B):
            s = gogame.init_state(size)
            for _ in range(int(rng.integers(0, 3 * size * size))):
                if gogame.game_ended(s):
                    break
                valid = np.flatnonzero(np.append(s[govars.INVD_CHNL].ravel(), 0) == 0)
                s = gogame.next_state(s, int(rng.choice(valid)))
            if gogame.game_ended(s) and b % 2:
                s = gogame.init_state(size)
            valid = np.flatnonzero(np.append(s[govars.INVD_CHNL].ravel(), 0) == 0)
            a = size * size if (b % 3 == 0 or gogame.game_ended(s)) else int(rng.choice(valid))
            states.append(s)
            actions.append(a)
        for canon in (False, True):
            nxt = np.stack([u8(gogame.next_state(s, a, canon)) for s, a in zip(states, actions)])
            out['n%d/next%s' % (size, '_canonical' if canon else '')] = nxt
        out['n%d/states' % size] = np.stack([u8(s) for s in states])
        out['n%d/actions' % size] = np.array(actions, dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, 'batch_passes.npz'), **out)
    print('batch_passes: %d arrays' % len(out))


def rollout(gogame, govars):
    out = {}
    for size, B, plies in ((5, 16, 80), (9, 16, 200), (19, 6, 700)):
        seed = 20260927
        rng0 = c_oracle.rng_seed(seed, B)
        states = np.zeros((B, 6, size, size), dtype=np.uint8)
        rng = rng0.copy()
        ref = [gogame.init_state(size) for _ in range(B)]
        n_resets = 0
        for t in range(plies):
            states, rng, last = c_oracle.batch_rollout(states, rng, 1, True)
            for b in range(B):
                if gogame.game_ended(ref[b]):
                    ref[b] = gogame.init_state(size)   # build-side auto-reset policy
                    n_resets += 1
                # the sampled action must be valid in the reference's eyes, then replay it there
                assert gogame.valid_moves(ref[b])[last[b]] == 1
                ref[b] = gogame.next_state(ref[b], int(last[b]))
                assert np.array_equal(u8(ref[b]), states[b]), (size, t, b)
        k = 'n%d/' % size
        out[k + 'seed'] = np.array(seed, dtype=np.uint64)
        out[k + 'plies'] = np.array(plies, dtype=np.int32)
        out[k + 'rng0'] = rng0
        out[k + 'rng_final'] = rng
        out[k + 'final_states'] = np.stack([u8(s) for s in ref])
        out[k + 'last_actions'] = last
        print('rollout n=%d: %d resets over %d plies x %d games' % (size, n_resets, plies, B))
    np.savez_compressed(os.path.join(HERE, 'rollout.npz'), **out)


def extras(gogame, govars, state_utils):
    rng = np.random.default_rng(17)
    out = {}
    # --- gogame.str (gym_go/gogame.py:407-468): empty, mid-game, passed, ended; stones on every edge / corner
    texts, tstates = [], []
    for size in (2, 5, 9, 19):
        s = gogame.init_state(size)
        tstates.append(u8(s)); texts.append(gogame.str(s))
        for ply in range(3 * size * size):
            if gogame.game_ended(s):
                break
            valid = np.flatnonzero(np.append(s[govars.INVD_CHNL].ravel(), 0) == 0)
            s = gogame.next_state(s, int(rng.choice(valid)))
            if ply % max(1, size * size // 3) == 0 or gogame.prev_player_passed(s):
                tstates.append(u8(s)); texts.append(gogame.str(s))
        tstates.append(u8(s)); texts.append(gogame.str(s))
    for i, (st, tx) in enumerate(zip(tstates, texts)):
        out['str/%d/state' % i] = st
        out['str/%d/text' % i] = np.array(tx)
    out['str/count'] = np.array(len(texts), dtype=np.int32)
    # --- gogame.all_symmetries (gym_go/gogame.py:362-382) on asymmetric images
    for j, (c, n) in enumerate(((6, 5), (3, 9), (1, 19), (6, 2))):
        img = rng.integers(0, 255, size=(c, n, n)).astype(np.uint8)
        out['sym/%d/image' % j] = img
        out['sym/%d/views' % j] = np.stack([np.ascontiguousarray(v) for v in gogame.all_symmetries(img)])
    out['sym/count'] = np.array(4, dtype=np.int32)
    # --- state_utils.update_pieces (gym_go/state_utils.py:159-180) on arbitrary positions
    cases = []
    for size in (2, 3, 5, 9, 19):
        for trial in range(40 if size < 19 else 24):
            dens = rng.uniform(0.3, 1.0)
            cells = rng.random((size, size))
            stone = cells < dens
            colour = rng.random((size, size)) < rng.uniform(0.25, 0.75)
            s = np.zeros((6, size, size))
            s[0][stone & colour] = 1
            s[1][stone & ~colour] = 1
            player = int(rng.integers(0, 2))
            mode = trial % 4
            if mode == 0:      # the reference's own call shape: neighbours of a stone of `player`
                own = np.argwhere(s[player] == 1)
                if len(own) == 0:
                    s[player, 0, 0] = 1; s[1 - player, 0, 0] = 0
                    own = np.array([[0, 0]])
                corner_bias = [p for p in own if (p[0] in (0, size - 1)) and (p[1] in (0, size - 1))]
                pt = corner_bias[0] if corner_bias and trial % 8 == 0 else own[rng.integers(len(own))]
                adj, _ = state_utils.adj_data(s, np.array(pt), player)
            elif mode == 1:    # neighbours of each corner in turn (whatever stands on the corner)
                cr = [(0, 0), (0, size - 1), (size - 1, 0), (size - 1, size - 1)][(trial // 4) % 4]
                adj, _ = state_utils.adj_data(s, np.array(cr), player)
            elif mode == 2:    # arbitrary locations, duplicates allowed, 1..4 of them
                k = int(rng.integers(1, 5))
                adj = rng.integers(0, size, size=(k, 2))
            else:              # a long list (more than four locations)
                k = int(rng.integers(5, 9))
                adj = rng.integers(0, size, size=(k, 2))
            before = u8(s)
            groups = state_utils.update_pieces(s, np.asarray(adj), player)
            killed = np.zeros((size, size), dtype=np.uint8)
            for g in groups:
                killed[g[:, 0], g[:, 1]] = 1
            cases.append((before, np.asarray(adj, dtype=np.int32).reshape(-1, 2), player, u8(s), killed, len(groups)))
    for i, (before, adj, player, after, killed, ng) in enumerate(cases):
        k = 'up/%d/' % i
        out[k + 'state'] = before
        out[k + 'adj'] = adj
        out[k + 'player'] = np.array(player, dtype=np.int32)
        out[k + 'after'] = after
        out[k + 'killed'] = killed
        out[k + 'groups'] = np.array(ng, dtype=np.int32)
    out['up/count'] = np.array(len(cases), dtype=np.int32)
    n_kill = sum(int(c[5] > 0) for c in cases)
    np.savez_compressed(os.path.join(HERE, 'extras.npz'), **out)
    print('extras: %d texts, 4 symmetry images, %d update_pieces cases (%d with captures)' % (len(texts), len(cases), n_kill))


def policy(gogame, govars):
    from sklearn import preprocessing
    rng = np.random.default_rng(23)
    out = {}
    cases = []
    for size in (2, 5, 9, 19):
        A = size * size + 1
        # a mid-game position of the reference itself: its plane 3 is the mask
        s = gogame.init_state(size)
        for _ in range(int(1.2 * size * size)):
            if gogame.game_ended(s):
                break
            valid = np.flatnonzero(np.append(s[govars.INVD_CHNL].ravel(), 0) == 0)
            s = gogame.next_state(s, int(rng.choice(valid)))
        if gogame.game_ended(s):
            s = gogame.init_state(size)
        invalid = np.append(s[govars.INVD_CHNL].ravel(), 0)
        for kind in ('uniform_valid', 'softmax', 'peaked', 'sparse'):
            if kind == 'uniform_valid':      # gogame.random_action's own weights
                w = 1 - invalid
            elif kind == 'softmax':
                z = rng.normal(0, 2.0, A)
                w = np.exp(z - z.max()) * (1 - invalid)
            elif kind == 'peaked':
                w = (rng.random(A) ** 8) * (1 - invalid)
                w[int(np.flatnonzero(1 - invalid)[0])] = 40.0
            else:
                w = np.where(rng.random(A) < 0.2, rng.random(A), 0.0) * (1 - invalid)
                w[A - 1] = 0.05
            w = w.astype(np.float32)
            p = preprocessing.normalize(w.astype(np.float64)[np.newaxis], norm='l1')[0]   # gym_go/gogame.py:391
            np.random.seed(1000 + len(cases))
            draws = np.array([gogame.random_weighted_action(w.astype(np.float64)) for _ in range(20000)])
            cases.append((size, kind))
            k = 'case/%d/' % (len(cases) - 1)
            out[k + 'size'] = np.array(size, dtype=np.int32)
            out[k + 'kind'] = np.array(kind)
            out[k + 'state'] = u8(s)
            out[k + 'weights'] = w
            out[k + 'p_reference'] = p
            out[k + 'hist_reference'] = np.bincount(draws, minlength=A).astype(np.int32)
        # gogame.random_action on the same state: its histogram
        np.random.seed(77 + size)
        ra = np.array([gogame.random_action(s) for _ in range(20000)])
        out['random_action/%d/state' % size] = u8(s)
        out['random_action/%d/hist_reference' % size] = np.bincount(ra, minlength=A).astype(np.int32)
    out['case/count'] = np.array(len(cases), dtype=np.int32)
    # the build's exact sampler on batches of reference-made states: (states, weights, rng) -> actions
    for size in (5, 9, 19):
        A = size * size + 1
        B = 64
        states = []
        for b in range(B):
            s = gogame.init_state(size)
            for _ in range(int(rng.integers(0, 2 * size * size))):
                if gogame.game_ended(s):
                    break
                valid = np.flatnonzero(np.append(s[govars.INVD_CHNL].ravel(), 0) == 0)
                s = gogame.next_state(s, int(rng.choice(valid)))
            states.append(u8(s))
        states = np.stack(states)
        w = (rng.random((B, A)) ** 3).astype(np.float32)
        w[::7] *= 1e-30
        w[3::11, : A // 2] = 0
        w[5] = 0                      # nothing playable has weight: -1
        w[6, :-1] = 0
        w[6, -1] = 1e-3               # only the pass
        w[9, 0] = -4.0                # a negative weight counts as 0
        rng0 = c_oracle.rng_seed(31 + size, B)
        acts, rng1 = c_oracle.batch_sample_weighted(states, w, rng0)
        k = 'exact/%d/' % size
        out[k + 'states'], out[k + 'weights'], out[k + 'rng0'], out[k + 'rng1'], out[k + 'actions'] = states, w, rng0, rng1, acts
    np.savez_compressed(os.path.join(HERE, 'policy.npz'), **out)
    print('policy: %d weight vectors x 20 000 reference draws, exact-sampler batches at 5 / 9 / 19' % len(cases))


def main():
    gym, gogame, govars, state_utils = refimport.load()
    jobs = {
        'scripted': lambda: scripted(gym, gogame),
        'random_games': lambda: random_games(gogame, govars),
        'children': lambda: children(gogame, govars),
        'batch_passes': lambda: batch_passes(gogame, govars),
        'rollout': lambda: rollout(gogame, govars),
        'extras': lambda: extras(gogame, govars, state_utils),
        'policy': lambda: policy(gogame, govars),
    }
    for name in (sys.argv[1:] or list(jobs)):
        jobs[name]()


if __name__ == '__main__':
    main()
