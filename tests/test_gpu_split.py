"""-m gpu: the persistent per-pair kernels on batches large enough for the resident grid split by wave age
(gg_kernels.hip age_split / gg_common.h pair_span: from two pairs per resident wave on, 12 288 - 16 384 boards on 256
CUs).  Which wave steps which pair must not matter: the whole batch in one call equals the same batch in chunks small
enough for the plain grid (pinned to the oracle by the other suites), bit for bit, odd batch sizes and a sub-sample
against the oracle included."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

CHUNK = 3000   # 1 500 pairs: below the split threshold of every device with >= 47 CUs


def _mixed(B, N, seed):
    """every game phase: stripe g of 16 has played g * N*N/9 plies, finished games are kept (auto_reset off on odd stripes)"""
    from gymgo_amd import gogame
    st = gogame.batch_init_state(B, N, device='cuda')
    rng = gogame.rng_seed(B, seed)
    ch = (B + 15) // 16
    for g in range(1, 16):
        lo, hi = g * ch, min(B, (g + 1) * ch)
        if lo < hi:
            gogame.batch_rollout(st[lo:hi], rng[lo:hi], g * (N * N // 9), g % 2 == 0)
    return st, rng


def _chunks(B):
    return [(lo, min(B, lo + CHUNK)) for lo in range(0, B, CHUNK)]


@pytest.mark.parametrize('N,B', [(19, 40001), (13, 36865), (9, 33000), (19, 65553), (13, 66001), (9, 70003)])
def test_whole_batch_equals_chunks(N, B):
    from gymgo_amd import gogame, state_utils
    st, rng = _mixed(B, N, 31 + N)
    cuts = _chunks(B)

    def both(fn_whole, fn_chunk):
        """fn_whole() -> tensors; fn_chunk(lo, hi) -> tensors of the slice; compared along dim 0"""
        whole = fn_whole()
        parts = [fn_chunk(lo, hi) for lo, hi in cuts]
        for i, w in enumerate(whole):
            got = torch.cat([p[i] for p in parts], 0)
            assert torch.equal(w, got), (N, B, i)
        return whole

    # gg_batch_invalid_mask (with a ko point on every third board), gg_batch_track_states
    ko = torch.full((B,), -1, dtype=torch.int32, device='cuda')
    ko[::3] = torch.arange(B, device='cuda', dtype=torch.int32)[::3] % (N * N)
    both(lambda: (gogame._invalid_mask_dev(st, ko),), lambda lo, hi: (gogame._invalid_mask_dev(st[lo:hi], ko[lo:hi]),))
    tracked, = both(lambda: (gogame.batch_track(st),), lambda lo, hi: (gogame.batch_track(st[lo:hi]),))
    assert torch.equal(gogame.batch_untrack(tracked), st)

    # gg_batch_env_step on byte planes and on packed boards: drawn moves, both reward methods, in place
    for method in ('real', 'heuristic'):
        for packed in (False, True):
            step = gogame.batch_env_step_packed if packed else gogame.batch_env_step
            base = gogame.batch_pack(st) if packed else st
            a, ra = base.clone(), rng.clone()
            b, rb = base.clone(), rng.clone()
            wa = step(a, None, ra, 5.5, method, True)
            wb = [step(b[lo:hi], None, rb[lo:hi], 5.5, method, True) for lo, hi in cuts]
            for i in range(4):
                assert torch.equal(wa[i], torch.cat([w[i] for w in wb], 0)), (N, method, packed, i)
            assert torch.equal(a, b) and torch.equal(ra, rb), (N, method, packed)
            if not packed and method == 'real':
                stepped, taken = a, wa[3]

    # ... and given moves (some illegal / out of range): the drawn ones of above, perturbed
    acts = taken.clone()
    acts[::7] = (acts[::7] * 5 + 3) % (N * N + 3) - 1
    for packed in (False, True):
        step = gogame.batch_env_step_packed if packed else gogame.batch_env_step
        base = gogame.batch_pack(st) if packed else st
        a, b = base.clone(), base.clone()
        wa = step(a, acts, None, 0.0, 'real', False)
        wb = [step(b[lo:hi], acts[lo:hi], None, 0.0, 'real', False) for lo, hi in cuts]
        for i in range(4):
            assert torch.equal(wa[i], torch.cat([w[i] for w in wb], 0)), (N, packed, i)
        assert torch.equal(a, b)

    # the one- and two-ply rollouts (per-ply kernel), packed next states, given move lists
    for plies in (1, 2):
        for packed in (False, True):
            roll = gogame.batch_rollout_packed if packed else gogame.batch_rollout
            base = gogame.batch_pack(st) if packed else st
            a, ra, b, rb = base.clone(), rng.clone(), base.clone(), rng.clone()
            la, lb = (torch.full((B,), -9, dtype=torch.int32, device='cuda') for _ in range(2))
            sa, sb = (torch.zeros(B, dtype=torch.int64, device='cuda') for _ in range(2))
            roll(a, ra, plies, True, la, sa)
            for lo, hi in cuts:
                roll(b[lo:hi], rb[lo:hi], plies, True, lb[lo:hi], sb[lo:hi])
            assert torch.equal(a, b) and torch.equal(ra, rb) and torch.equal(la, lb) and torch.equal(sa, sb), (N, plies, packed)
    pk = gogame.batch_pack(st)
    both(lambda: gogame.batch_next_states_packed(pk, acts, check=False),
         lambda lo, hi: gogame.batch_next_states_packed(pk[lo:hi], acts[lo:hi], check=False))
    both(lambda: gogame.batch_next_states(st, acts, check=False),
         lambda lo, hi: gogame.batch_next_states(st[lo:hi], acts[lo:hi], check=False))
    moves = torch.stack([acts, taken, acts], 1).contiguous()
    for packed in (False, True):
        base = gogame.batch_pack(st) if packed else st
        a, b = base.clone(), base.clone()
        pa = gogame.batch_play_moves(a, moves)
        pb = torch.cat([gogame.batch_play_moves(b[lo:hi], moves[lo:hi]) for lo, hi in cuts], 0)
        assert torch.equal(a, b) and torch.equal(pa, pb), (N, packed)

    # a strided sub-sample of the big env step against the oracle (gym_go/envs/go_env.py:49-76 via gogame.next_state)
    from oracle import c_oracle
    idx = np.arange(0, B, 97)
    before = st.cpu().numpy()[idx]
    tk = taken.cpu().numpy()[idx]
    after = stepped.cpu().numpy()[idx]
    for j in range(len(idx)):
        s0 = before[j]
        if s0[5].any():   # a finished game was reset first (auto_reset)
            s0 = np.zeros_like(s0)
        want = c_oracle.batch_next_states(s0[None], tk[j:j + 1], False)[0][0]
        assert np.array_equal(after[j], want), (N, int(idx[j]))


def test_first_use_inside_a_hipgraph_capture():
    """The host side asks the runtime for a kernel's occupancy the first time it launches it (age_split); that first
    time may be inside a stream capture (a GoVecEnv step loop captured as a hipGraph): a fresh process captures
    gg_batch_env_step on a batch large enough for the split, replays it, and gets the eager result."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = '''
import sys, torch
sys.path.insert(0, %r)
from gymgo_amd import gogame
B, N = 16385, 19
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 7)
out = (torch.empty(B, dtype=torch.float32, device='cuda'), torch.empty(B, dtype=torch.uint8, device='cuda'),
       torch.empty(B, dtype=torch.int32, device='cuda'), torch.empty(B, dtype=torch.int32, device='cuda'))
ref, rref = st.clone(), rng.clone()
g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        gogame.batch_env_step(st, None, rng, 7.5, 'real', True, out=out)
for _ in range(5): g.replay()
torch.cuda.synchronize()
for _ in range(5): gogame.batch_env_step(ref, None, rref, 7.5, 'real', True)
assert torch.equal(st, ref) and torch.equal(rng, rref)
print('SAME')
''' % root
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'SAME' in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize('N,B', [(19, 53001), (13, 50003), (9, 70001)])
def test_next_states_sixteen_boards_per_wave_equals_two_board_kernel(N, B):
    """gg_batch_next_states takes the sixteen-boards-per-wave kernel (k_next_states16, gg_ns16.h) from one group per
    resident wave on (49 152 boards on 256 CUs) for 9x9 / 13x13 / 19x19: the whole batch - every game phase, legal moves,
    passes, moves on occupied / suicide / ko points, out-of-range actions, finished games, both `canonical` settings,
    a ragged last group - equals the same batch in chunks that take the two-board kernel, and a sub-sample the oracle
    (gym_go/gogame.py:34-87)."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    st, rng = _mixed(B, N, 77 + N)
    acts = gogame.batch_sample_actions(st, rng)
    gen = torch.Generator(device='cuda'); gen.manual_seed(N)
    wild = torch.rand(B, device='cuda', generator=gen) < 0.3
    rnd = torch.randint(-2, N * N + 3, (B,), device='cuda', generator=gen, dtype=torch.int32)
    acts = torch.where(wild, rnd, acts)
    for canon in (False, True):
        out, status = gogame.batch_next_states(st, acts, canonical=canon, check=False)
        parts = [gogame.batch_next_states(st[lo:hi], acts[lo:hi], canonical=canon, check=False) for lo, hi in _chunks(B)]
        assert torch.equal(status, torch.cat([p[1] for p in parts], 0)), (N, canon)
        assert torch.equal(out, torch.cat([p[0] for p in parts], 0)), (N, canon)
        assert int((status != 0).sum()) > B // 20   # the refused moves are there
        idx = np.arange(0, B, 131)
        ok = idx[status.cpu().numpy()[idx] == 0]
        host = st.cpu().numpy()[ok]
        want = c_oracle.batch_next_states(host, acts.cpu().numpy()[ok], canon)[0]
        assert np.array_equal(out.cpu().numpy()[ok], want), (N, canon)
    # unaligned views (a batch that starts in the middle of another: the group's byte range starts at any address mod 16)
    sub = st[7:7 + 50000 + (N == 9) * 20000]
    o1, s1 = gogame.batch_next_states(sub, acts[7:7 + len(sub)], check=False)
    o2, s2 = gogame.batch_next_states(sub.clone(), acts[7:7 + len(sub)].clone(), check=False)
    assert torch.equal(o1, o2) and torch.equal(s1, s2)


@pytest.mark.parametrize('N,B', [(19, 65553), (13, 66001), (9, 70003)])
def test_invalid_mask_sixteen_boards_per_wave(N, B):
    """gg_batch_invalid_mask on big batches (k_invalid_mask16): equal to the same batch in chunks that take the two-board
    kernel, with and without ko points, every game phase, finished games included; a sub-sample against the oracle's
    state_utils.compute_invalid_moves (gym_go/state_utils.py:24-83)."""
    from gymgo_amd import gogame
    from oracle import c_oracle
    st, _ = _mixed(B, N, 5 + N)
    ko = torch.full((B,), -1, dtype=torch.int32, device='cuda')
    ko[::3] = torch.arange(B, device='cuda', dtype=torch.int32)[::3] % (N * N)
    for kk in (None, ko):
        whole = gogame._invalid_mask_dev(st, kk)
        parts = torch.cat([gogame._invalid_mask_dev(st[lo:hi], None if kk is None else kk[lo:hi]) for lo, hi in _chunks(B)], 0)
        assert torch.equal(whole, parts), (N, kk is None)
    idx = np.arange(0, B, 257)
    host = st.cpu().numpy()[idx]
    got = gogame._invalid_mask_dev(st, None).cpu().numpy()[idx]
    for j in range(len(idx)):
        want = c_oracle.compute_invalid_moves(host[j], 1 - int(host[j][2, 0, 0]), -1)   # (the mask is for the opponent of `player`)
        assert np.array_equal(got[j], want), (N, int(idx[j]))
    sub = st[5:5 + 65536 + 16]
    assert torch.equal(gogame._invalid_mask_dev(sub, None), gogame._invalid_mask_dev(sub.clone(), None))


@pytest.mark.parametrize('N', [19, 13, 9])
def test_sixteen_board_kernels_on_snakes_spirals_and_combs(N):
    """The class-major kernels on the shapes that stress the flood (one-stone-wide spirals, serpentines, combs with
    opponent stones in the corridors: many sweep rounds, big captures): 24 synthetic layouts tiled to 66 000 boards, every
    board with its own move - gg_batch_next_states, gg_batch_env_step (drawn moves) and gg_batch_invalid_mask on the
    whole batch equal the same batch in chunks on the two-board kernels (which test_gpu_adversarial.py pins to the
    oracle), and the first 24 x 8 boards the oracle directly."""
    from test_gpu_adversarial import consistent_boards
    from gymgo_amd import gogame
    from oracle import c_oracle
    base = consistent_boards(N)
    d0 = torch.from_numpy(base).cuda()
    base[:, 3] = gogame._invalid_mask_dev(d0, None).cpu().numpy()      # plane 3 of a legal state
    B = 66000
    reps = (B + len(base) - 1) // len(base)
    st = torch.from_numpy(np.tile(base, (reps, 1, 1, 1))[:B].copy()).cuda()
    rng = gogame.rng_seed(B, 1234 + N)
    acts = gogame.batch_sample_actions(st, rng)                        # a different legal move per copy
    acts[::11] = (acts[::11] * 7 + 1) % (N * N + 2) - 1                # ... and some that are not
    out, status = gogame.batch_next_states(st, acts, check=False)
    parts = [gogame.batch_next_states(st[lo:hi], acts[lo:hi], check=False) for lo, hi in _chunks(B)]
    assert torch.equal(out, torch.cat([p[0] for p in parts], 0)) and torch.equal(status, torch.cat([p[1] for p in parts], 0))
    head = 24 * 8
    ok = np.flatnonzero(status[:head].cpu().numpy() == 0)
    want = c_oracle.batch_next_states(st[:head].cpu().numpy()[ok], acts[:head].cpu().numpy()[ok], False)[0]
    assert np.array_equal(out[:head].cpu().numpy()[ok], want)
    a, ra, b, rb = st.clone(), rng.clone(), st.clone(), rng.clone()
    wa = gogame.batch_env_step(a, None, ra, 6.5, 'heuristic', True)
    wb = [gogame.batch_env_step(b[lo:hi], None, rb[lo:hi], 6.5, 'heuristic', True) for lo, hi in _chunks(B)]
    for i in range(4):
        assert torch.equal(wa[i], torch.cat([w[i] for w in wb], 0)), (N, i)
    assert torch.equal(a, b) and torch.equal(ra, rb)
    m1 = gogame._invalid_mask_dev(out, None)
    m2 = torch.cat([gogame._invalid_mask_dev(out[lo:hi], None) for lo, hi in _chunks(B)], 0)
    assert torch.equal(m1, m2)
