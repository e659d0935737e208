"""Un-padded children on the device: gogame.children(state, canonical, padded=False) (gym_go/gogame.py:175-180 - the un-padded
result is what the reference computes first, :179; GoEnv.children forwards the flag, gym_go/envs/go_env.py:105-109) and its
batch form (gg_batch_children_offsets + gg_batch_children_compact) against the reference-recorded golden children, the C
oracle and the padded entry point."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SEED = 20260927


def _valid(states):
    """The reference's valid_moves per state (gym_go/gogame.py:153-161): plane 3 clear + the pass; everything once ended."""
    B, _, N, _ = states.shape
    v = np.concatenate([states[:, 3].reshape(B, -1) == 0, np.ones((B, 1), bool)], axis=1)
    v[states[:, 5, 0, 0] != 0] = True
    return v


def test_children_unpadded_golden(golden):
    """tests/golden/children.npz (recorded from the reference's children(padded=True)): the un-padded form is the padded one
    without the slots valid_moves() drops - single state (NumPy in / out) and as a batch of one."""
    from gymgo_amd import gogame as gg
    z = golden('children')
    keys = sorted({k.split('/')[0] for k in z.files})
    assert keys
    for key in keys:
        st = z[key + '/state'].astype(np.uint8)
        for canon, name in ((False, '/children'), (True, '/children_canonical')):
            want = z[key + name].astype(np.uint8)[_valid(st[None])[0]]
            got = gg.children(st, canonical=canon, padded=False)
            assert got.shape == want.shape and np.array_equal(got.astype(np.uint8), want), (key, canon)
            kids, offs = gg.batch_children(torch.from_numpy(st[None]).cuda(), canonical=canon, padded=False)
            assert offs.tolist() == [0, len(want)] and np.array_equal(kids.cpu().numpy(), want)


@pytest.mark.parametrize('N,B', [(19, 8192), (9, 3000), (13, 1001), (5, 257), (2, 5), (19, 1), (5, 16384), (5, 16385), (4, 40001)])   # (the last: the tiled scan of the offsets)
def test_batch_children_compact_vs_padded_and_oracle(N, B):
    """Mid-game parents (a few finished games among them - every action is "valid" there, gogame.py:155-156): offsets ==
    the exclusive scan of the valid counts; the concatenation == the padded kernel's kept slots for EVERY parent; == the C
    oracle's for a sub-sample; canonical both ways; a caller-owned upper-bound buffer gives the same bytes."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    A = N * N + 1
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, SEED + N, 0, 'cuda')
    q = max(1, B // 4)
    for g in range(4):
        lo, hi = g * q, (B if g == 3 else (g + 1) * q)
        if lo < hi:
            gogame.batch_rollout(st[lo:hi], rng[lo:hi], (N * N) // 6 + (N * N * g) // 4, auto_reset=False)
    host = st.cpu().numpy()
    valid = _valid(host)
    counts = valid.sum(axis=1)
    want_offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    for canon in (False, True):
        kids, offs = gogame.batch_children(st, canonical=canon, padded=False)
        assert np.array_equal(offs.cpu().numpy(), want_offs)
        assert kids.shape == (int(want_offs[-1]), 6, N, N)
        padded = gogame.batch_children(st, canonical=canon, padded=True)
        assert torch.equal(kids, padded[torch.from_numpy(valid).cuda()]), (N, canon)
        del padded
        sub = np.arange(0, B, max(1, B // 128))[:128]
        ora = c_oracle.batch_children_mt(host[sub], canon)
        for j, b in enumerate(sub):
            assert np.array_equal(kids[int(want_offs[b]):int(want_offs[b + 1])].cpu().numpy(), ora[j][valid[b]]), (N, canon, int(b))
        if not canon:
            # caller-owned buffers: an upper-bound children buffer with a guard value behind the total, offsets reused
            buf = torch.full((B * A + 3, 6, N, N), 0x5A, dtype=torch.uint8, device='cuda')
            offs2 = torch.empty(B + 1, dtype=torch.int32, device='cuda')
            k2, o2 = gogame.batch_children(st, canonical=False, padded=False, out=buf, offsets=offs2)
            total = int(want_offs[-1])
            assert o2.data_ptr() == offs2.data_ptr() and torch.equal(o2, offs)
            assert torch.equal(buf[:total], kids) and bool((buf[total:] == 0x5A).all())
            del buf
        del kids


def test_children_compact_single_state_forms():
    """gogame.children(padded=False) / GoEnv.children(padded=False): NumPy in -> NumPy out like the reference, == padded[valid]."""
    from gymgo_amd import gogame
    from gymgo_amd.envs import GoEnv
    env = GoEnv(size=7, komi=0)
    env.reset()
    rs = np.random.default_rng(3)
    for _ in range(25):
        env.step(int(rs.choice(np.flatnonzero(env.valid_moves()))))
        if env.done:
            break
    for canon in (False, True):
        a = env.children(canonical=canon, padded=True)
        b = env.children(canonical=canon, padded=False)
        assert isinstance(b, np.ndarray) and b.dtype == a.dtype
        assert np.array_equal(b, a[gogame.valid_moves(env.state_) == 1])


def test_children_compact_argument_checks():
    from gymgo_amd import _lib
    L = _lib.lib()
    st = torch.zeros((2, 6, 5, 5), dtype=torch.uint8, device='cuda')
    offs = torch.empty(3, dtype=torch.int32, device='cuda')
    order = torch.empty(2, dtype=torch.int32, device='cuda')
    assert L.gg_batch_children_offsets(st.data_ptr(), None, None, 2, 5, None) == -2
    assert L.gg_batch_children_offsets(None, offs.data_ptr(), None, 2, 5, None) == -2
    assert L.gg_batch_children_offsets(st.data_ptr(), offs.data_ptr(), None, 2, 1, None) == -1
    assert L.gg_batch_children_offsets(st.data_ptr(), offs.data_ptr(), None, 1 << 40, 19, None) == -1     # B (N^2+1) overflows int32
    assert L.gg_batch_children_compact(st.data_ptr(), None, None, st.data_ptr(), 2, 5, 0, None) == -2
    assert L.gg_batch_children_offsets(st.data_ptr(), offs.data_ptr(), None, 0, 5, None) == 0
    torch.cuda.synchronize()
    assert int(offs[0]) == 0
    assert L.gg_batch_children_offsets(st.data_ptr(), offs.data_ptr(), order.data_ptr(), 2, 5, None) == 0
    torch.cuda.synchronize()
    assert offs.tolist() == [0, 26, 52] and sorted(order.tolist()) == [0, 1]


def test_children_compact_work_order_is_a_permutation_and_does_not_matter():
    """`order` (the parents by falling child count: heaviest first) decides which work item a parent is, never what is written:
    the library's order, no order (NULL: index order) and a random permutation give identical bytes."""
    from gymgo_amd import _lib, gogame
    L = _lib.lib()
    B, N = 3001, 9
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, 5, 0, 'cuda')
    for g in range(3):
        gogame.batch_rollout(st[g * 1000:(g + 1) * 1000], rng[g * 1000:(g + 1) * 1000], 10 + 35 * g, auto_reset=False)
    offs = torch.empty(B + 1, dtype=torch.int32, device='cuda')
    order = torch.empty(B, dtype=torch.int32, device='cuda')
    assert L.gg_batch_children_offsets(st.data_ptr(), offs.data_ptr(), order.data_ptr(), B, N, None) == 0
    torch.cuda.synchronize()
    counts = (offs[1:] - offs[:-1]).cpu().numpy()
    o = order.cpu().numpy()
    assert sorted(o.tolist()) == list(range(B))
    assert (np.diff(counts[o]) <= 0).all()                      # falling child count
    total = int(offs[B])
    outs = []
    for od in (order, None, torch.randperm(B, device='cuda').to(torch.int32)):
        buf = torch.full((total, 6, N, N), 0x77, dtype=torch.uint8, device='cuda')
        assert L.gg_batch_children_compact(st.data_ptr(), offs.data_ptr(), None if od is None else od.data_ptr(), buf.data_ptr(),
                                           B, N, 0, None) == 0
        torch.cuda.synchronize()
        outs.append(buf)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_children_compact_refuses_short_buffers_and_counts_bit_0_of_plane_3():
    """caller-owned `out` / `offsets` / `order` that are too short are refused on the host (they were silent
    out-of-bounds device writes), and a plane-3 byte such as 2 (a caller-edited state) counts by bit 0 in BOTH the offsets and
    the expansion, so a parent never emits more children than its slot holds."""
    from gymgo_amd import gogame
    B, N = 40, 5
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, 11, 0, 'cuda')
    gogame.batch_rollout(st, rng, 9, auto_reset=False)
    full, offs = gogame._children_compact_dev(st, False)
    total = int(offs[B])
    with pytest.raises(ValueError):
        gogame._children_compact_dev(st, False, out=torch.empty((total - 1, 6, N, N), dtype=torch.uint8, device='cuda'))
    with pytest.raises(ValueError):
        gogame._children_compact_dev(st, False, offsets=torch.empty(B, dtype=torch.int32, device='cuda'))
    with pytest.raises(ValueError):
        gogame._children_offsets_dev(st, order=torch.empty(B - 1, dtype=torch.int32, device='cuda'))
    exact = torch.full((total, 6, N, N), 7, dtype=torch.uint8, device='cuda')     # exactly the total: accepted after one host read
    got, _ = gogame._children_compact_dev(st, False, out=exact)
    assert torch.equal(got, full)
    # plane-3 bytes of 2: bit 0 clear = a kept point, for the count and for the expansion alike
    odd = st.clone()
    odd[:, 3] = odd[:, 3] * 3          # set points become 3 (bit 0 set), clear ones stay 0
    odd[:, 3][st[:, 3] == 0] = 2       # ... and the clear ones become 2 (bit 0 clear)
    guard = torch.full((total + 8, 6, N, N), 9, dtype=torch.uint8, device='cuda')
    got2, offs2 = gogame._children_compact_dev(odd, False, out=guard[:total])
    torch.cuda.synchronize()
    assert torch.equal(offs2, offs) and bool((guard[total:] == 9).all())
    # (plane 3 of a child is recomputed by the expansion; planes 0 / 1 and the flags come from the move)
    assert torch.equal(got2[:, :2], full[:, :2])
