"""-m gpu: shapes that stress the flood (snakes, spirals, combs: many sweep rounds) and BASELINE-size property tests."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def spiral(n):
    """0/1 mask of a one-cell-wide spiral corridor wall (a single snake group ~ n*n/2 stones)."""
    m = np.zeros((n, n), np.uint8)
    top, left, bot, right = 0, 0, n - 1, n - 1
    while top <= bot and left <= right:
        m[top, left:right + 1] = 1
        m[top:bot + 1, right] = 1
        if bot > top:
            m[bot, left:right + 1] = 1
        if right > left:
            m[top + 2:bot + 1, left] = 1
        top, left, bot, right = top + 2, left + 2, bot - 2, right - 2
        if top <= bot and left - 1 >= 0:
            m[top, left - 1] = 1     # connect to the next ring
    return m


def serpentine(n, vertical=False):
    m = np.zeros((n, n), np.uint8)
    for r in range(0, n, 2):
        m[r, :] = 1
        if r + 1 < n:
            m[r + 1, (n - 1) if (r // 2) % 2 == 0 else 0] = 1
    return m.T.copy() if vertical else m


def comb(n):
    m = np.zeros((n, n), np.uint8)
    m[0, :] = 1
    m[:, ::2] = 1
    return m


def boards(n):
    shapes = [spiral(n), serpentine(n), serpentine(n, True), comb(n), comb(n)[::-1].copy(), spiral(n)[:, ::-1].copy()]
    out = []
    for s in shapes:
        for colour in (0, 1):
            for turn in (0, 1):
                st = np.zeros((6, n, n), np.uint8)
                st[colour] = s
                # sprinkle opponent stones into some of the corridors (creates ataris / captures next to the snake)
                free = np.argwhere(s == 0)
                rng = np.random.default_rng(n * 7 + colour * 3 + turn)
                pick = free[rng.permutation(len(free))[:len(free) // 3]]
                st[1 - colour, pick[:, 0], pick[:, 1]] = 1
                st[2] = turn
                out.append(st)
    return np.stack(out)


@pytest.mark.parametrize('n', [9, 13, 19])
def test_snake_groups_vs_oracle(n):
    """Positions are synthetic (not reachable by play), so zero-liberty groups are removed first by stepping a
    pass through the ORACLE-consistent route: we only compare functions that are defined on any stone layout."""
    from gymgo_amd import gogame, state_utils
    from oracle import c_oracle
    st = consistent_boards(n)
    d = dev(st)
    # 1. invalid mask for the side to move (the flood-heavy part)
    got = state_utils.batch_compute_invalid_moves(d, None, None).cpu().numpy()
    for i in range(len(st)):
        want = c_oracle.compute_invalid_moves(st[i], 1 - int(st[i, 2, 0, 0]))
        assert np.array_equal(got[i], want), (n, i)
    st[:, 3] = got
    d = dev(st)
    # 2. areas
    b, w = gogame.batch_areas(d)
    ob, ow = c_oracle.batch_areas(st)
    assert np.array_equal(b.cpu().numpy(), ob) and np.array_equal(w.cpu().numpy(), ow)
    # 3. every legal move of every board (children, both canonical settings) and a pass
    for canon in (False, True):
        kids = gogame.batch_children(d, canonical=canon).cpu().numpy()
        assert np.array_equal(kids, c_oracle.batch_children(st, canon)), (n, canon)
    # 4. fused rollout from these positions (v2 atari carry-over across plies, captures of big snakes)
    rng = gogame.rng_seed(len(st), 5)
    rng_np = c_oracle.rng_seed(5, len(st))
    roll = d.clone()
    gogame.batch_rollout(roll, rng, 40, True)
    want, _, _ = c_oracle.batch_rollout(st, rng_np, 40, True)
    assert np.array_equal(roll.cpu().numpy(), want), n


def consistent_boards(n):
    """boards(n) with the stones of liberty-less groups dropped (both colours): layouts every function is defined on"""
    st = boards(n)
    for i in range(len(st)):
        for colour in (0, 1):
            lab_dead = []
            s = st[i]
            empt = (s[0] + s[1]) == 0
            # flood per group on the host (tiny): remove liberty-less groups
            seen = np.zeros((n, n), bool)
            for r0, c0 in np.argwhere(s[colour] == 1):
                if seen[r0, c0]:
                    continue
                stack, cells, lib = [(r0, c0)], [], False
                seen[r0, c0] = True
                while stack:
                    r, c = stack.pop()
                    cells.append((r, c))
                    for dr, dc in ((-1, 0), (1, 0), (0, -1), (0, 1)):
                        rr, cc = r + dr, c + dc
                        if 0 <= rr < n and 0 <= cc < n:
                            if empt[rr, cc]:
                                lib = True
                            elif s[colour, rr, cc] and not seen[rr, cc]:
                                seen[rr, cc] = True
                                stack.append((rr, cc))
                if not lib:
                    lab_dead += cells
            for r, c in lab_dead:
                st[i, colour, r, c] = 0
    return st


def test_full_size_properties_and_shard_invariance():
    """BASELINE config 3 size (19x19, 65 536 games), size-independent PROPERTIES of the HIP path (the oracle comparison
    at this size is tests/test_gpu_configs.py): determinism, shard invariance (the multi-GPU decomposition),
    structural invariants of every state, plane 3 == a fresh analysis up to one ko point, and a replay of a
    sub-sample through the oracle."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    B, N, plies = 65536, 19, 96
    seed = 20260927

    def run(first, count):
        st = gogame.batch_init_state(count, N, device='cuda')
        rng = gogame.rng_seed(count, seed, first)
        steps = torch.zeros(count, dtype=torch.int64, device='cuda')
        gogame.batch_rollout(st, rng, plies, True, None, steps)
        return st, steps

    whole, steps = run(0, B)
    assert int(steps.min()) == plies and int(steps.max()) == plies
    again, _ = run(0, B)
    assert torch.equal(whole, again)                                     # deterministic
    parts = [run(f, c)[0] for f, c in ((0, 8192), (8192, 8192), (16384, 49152))]
    assert torch.equal(whole, torch.cat(parts))                          # shard-invariant
    s = whole
    assert int((s[:, 0] & s[:, 1]).sum()) == 0                           # a point holds one stone at most
    assert bool((s <= 1).all())
    for p in (2, 4, 5):                                                  # uniform planes
        assert bool((s[:, p].amax(dim=(1, 2)) == s[:, p].amin(dim=(1, 2))).all())
    occ = s[:, 0] | s[:, 1]
    assert bool(((s[:, 3] & occ) == occ).all())                          # occupied points are invalid
    b, w = gogame.batch_areas(s)
    assert bool(((b + w) <= N * N).all()) and bool((b >= s[:, 0].sum(dim=(1, 2))).all())
    canon = gogame.batch_canonical_form(s)
    assert torch.equal(canon, gogame.batch_canonical_form(canon))        # idempotent
    # plane 3 is what a fresh analysis gives, except for (at most one) ko point per game
    from gymgo_amd import state_utils
    fresh = state_utils.batch_compute_invalid_moves(s, None, None)
    extra = (s[:, 3].to(torch.int16) - fresh.to(torch.int16))
    assert int(extra.min()) >= 0 and int(extra.sum(dim=(1, 2)).max()) <= 1
    # sub-sample replayed by the oracle from the empty board with the same generator
    idx = np.arange(0, B, 129)[:400]
    rng_np = np.array([c_oracle.lib().gg_oracle_rng_seed(seed, int(i)) for i in idx], dtype=np.uint64)
    want, _, _ = c_oracle.batch_rollout(np.zeros((len(idx), 6, N, N), np.uint8), rng_np, plies, True)
    assert np.array_equal(s[torch.as_tensor(idx, device='cuda')].cpu().numpy(), want)


def _sym(t, k):
    """The k-th of the 8 board symmetries on the last two dims of a tensor: k & 3 quarter turns, then a flip if k & 4."""
    t = torch.rot90(t, k & 3, dims=(-2, -1))
    return torch.flip(t, dims=(-1,)) if k & 4 else t


def test_full_size_symmetry_equivariance():
    """BASELINE config 3 size, a property that needs no oracle: the rules of Go commute with the 8 symmetries of the
    board.  For 65 536 mid-game boards of every phase and one sampled legal move each:
    next_state(sym(s), sym(a)) == sym(next_state(s, a)) for the byte-plane kernel AND for the workspace / multi-ply path,
    areas are invariant, and the padded children of a sub-batch are the permuted, transformed children."""
    from gymgo_amd import gogame
    B, N = 65536, 19
    P = N * N
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, 77)
    for g in range(16):
        gogame.batch_rollout(st[g * 4096:(g + 1) * 4096], rng[g * 4096:(g + 1) * 4096], 20 + 24 * g, auto_reset=False)
    gogame.batch_reset_finished(st)
    acts = gogame.batch_sample_actions(st, rng)
    base, status = gogame.batch_next_states(st, acts, check=False)
    assert int(status.sum()) == 0
    b0, w0 = gogame.batch_areas(base)
    # where each point goes under symmetry k: transform the index grid itself
    grid = torch.arange(P, dtype=torch.int32, device='cuda').reshape(N, N)
    ws = gogame.next_states_workspace(B, N, 'cuda')
    out = torch.empty_like(st)
    st_flag = torch.empty(B, dtype=torch.int32, device='cuda')
    for k in range(1, 8):
        moved = _sym(grid, k).reshape(-1)                  # moved[j] = the old index now at position j
        where = torch.empty(P + 1, dtype=torch.int32, device='cuda')
        where[moved.long()] = torch.arange(P, dtype=torch.int32, device='cuda')
        where[P] = P                                         # the pass stays the pass
        s_k = _sym(st, k).contiguous()
        a_k = where[acts.long()].contiguous()
        got, status = gogame.batch_next_states(s_k, a_k, check=False)
        assert int(status.sum()) == 0
        want = _sym(base, k)
        assert torch.equal(got, want), k
        # the same through the multi-ply kernel's one-ply path (every board misses the workspace: analysed from scratch
        # at load, then stepped by the quad-per-board ply)
        gogame.batch_next_states(s_k, a_k, check=False, out=out, status=st_flag, workspace=ws)
        assert torch.equal(out, want) and int(st_flag.sum()) == 0, ('workspace', k)
        bk, wk = gogame.batch_areas(got)
        assert torch.equal(bk, b0) and torch.equal(wk, w0), ('areas', k)
        if k in (3, 6):                                      # children of a sub-batch: slots permuted, boards transformed
            sub = slice(0, 65536, 257)
            kids = gogame.batch_children(st[sub].contiguous())
            kids_k = gogame.batch_children(s_k[sub].contiguous())
            perm = torch.cat([moved.long(), torch.tensor([P], device='cuda')])     # slot j of the transformed parent = old slot perm[j]
            assert torch.equal(kids_k, _sym(kids[:, perm], k)), ('children', k)


@pytest.mark.parametrize('N,B,plies', [(19, 96, (30, 120, 250, 330)), (13, 64, (20, 90, 150)), (9, 128, (10, 40, 70)),
                                        (5, 64, (6, 14, 22)), (3, 32, (3, 7)), (2, 16, (1, 3))])
def test_children_many_positions(N, B, plies):
    """gg_batch_children (incremental kernel: per-parent liberty counts + per-point group floods) on positions of
    every game phase - captures, ko, snapbacks, merges - both canonical settings, against the oracle
    (gym_go/gogame.py:175-186)."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, 1234 + N)
    done = 0
    for p in plies:
        gogame.batch_rollout(st, rng, p - done, auto_reset=False)
        done = p
        host = st.cpu().numpy()
        live = host[:, 5, 0, 0] == 0      # gogame.children of a finished game is undefined in the reference
        if not live.any():
            continue
        d = st[torch.from_numpy(live).cuda()].contiguous()
        for canon in (False, True):
            kids = gogame.batch_children(d, canonical=canon).cpu().numpy()
            want = c_oracle.batch_children(host[live], canon)
            assert np.array_equal(kids, want), (N, p, canon, np.argwhere((kids != want).reshape(len(kids), N * N + 1, -1).any(-1))[:5])


def test_children_single_chunk_path():
    """B large enough that every wave expands all N*N+1 slots of its parent (chunks == 1)."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    B, N = 4608, 9
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, 99)
    gogame.batch_rollout(st, rng, 45, auto_reset=True)
    host = st.cpu().numpy()
    live = host[:, 5, 0, 0] == 0
    d = st[torch.from_numpy(live).cuda()].contiguous()
    kids = gogame.batch_children(d, canonical=False).cpu().numpy()
    assert np.array_equal(kids, c_oracle.batch_children(host[live], False))


@pytest.mark.parametrize('layout', ['tracked', 'bytes'])
def test_env_step_config3_size_matches_fused_rollout(layout):
    """65 536 x 19x19 (BASELINE config 3): K fused GoEnv.step launches with on-device sampling walk exactly the
    trajectory of one K-ply gg_batch_rollout launch (same generator), rewards/dones consistent with the states."""
    from gymgo_amd import gogame
    from gymgo_amd.envs import GoVecEnv
    B, N, K = 65536, 19, 24
    env = GoVecEnv(B, N, komi=7.5, reward_method='heuristic', seed=11, layout=layout)
    env.rollout(200)
    ref = env.states.clone()
    ref_rng = env.rng.clone()
    for _ in range(K):
        states, rewards, dones, status = env.step()
        assert int(status.sum()) == 0
    gogame.batch_rollout(ref, ref_rng, K, True)
    assert torch.equal(states, ref) and torch.equal(env.rng, ref_rng)
    b, w = gogame.batch_areas(states)
    margin = (b - w).float() - 7.5
    over = states[:, 5, 0, 0] == 1
    want = torch.where(over, torch.where(margin > 0, 361.0, -361.0), margin)
    assert torch.equal(rewards, want) and torch.equal(dones, states[:, 5, 0, 0])
