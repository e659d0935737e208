"""Developer soak (not collected by pytest; run on the GPU box): long uniform-random self-play through the per-ply and
fused kernels, replayed move by move through the oracle, plus children of positions from every game phase."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gymgo_amd import gogame
from oracle import c_oracle

def soak(N, B, plies, seed):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, seed)
    pk = gogame.batch_pack(st); prng = rng.clone()
    want = np.zeros((B, 6, N, N), np.uint8); orng = c_oracle.rng_seed(seed, B)
    la = torch.empty(B, dtype=torch.int32, device='cuda')
    out = None
    t = 0
    while t < plies:
        k = int(np.random.default_rng(t).integers(1, 9))
        if k % 3 == 0:
            k += 8
            gogame.batch_rollout(st, rng, k, True, la, None)
            gogame.batch_rollout_packed(pk, prng, k, True)
        elif k % 3 == 1:
            for _ in range(k):
                out = gogame.batch_env_step(st, None, rng, 6.5, 'heuristic', True, out=out)
                gogame.batch_env_step_packed(pk, None, prng, 6.5, 'real', True)
        else:
            for _ in range(k):
                gogame.batch_reset_finished(st)
                a = gogame.batch_sample_actions(st, rng)
                st, status = gogame.batch_next_states(st, a, check=False)
                assert int(status.sum()) == 0
                gogame.batch_rollout_packed(pk, prng, 1, True)
        want, orng, _ = c_oracle.batch_rollout(want, orng, k, True)
        t += k
        if t % 50 < k:
            assert np.array_equal(st.cpu().numpy(), want), (N, t)
            assert torch.equal(gogame.batch_unpack(pk, N), st), (N, t)
    assert np.array_equal(st.cpu().numpy(), want) and np.array_equal(rng.cpu().numpy().view(np.uint64), orng)
    print('soak %dx%d: %d games x %d plies bit-exact (mixed fused / env-step / next_states / packed kernels)' % (N, N, B, t), flush=True)

def soak_multi_ply(N, B, plies, seed, every=16):
    """The multi-ply kernel (16 boards per wave, classes carried in registers) at its own dispatch sizes: byte-plane,
    packed and tracked boards walk the same trajectory through launches of random length, the tracked env step and
    one-ply tracked launches in between; every `every`-th game is replayed by the oracle; class rows stay exact."""
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, seed)
    pk = gogame.batch_pack(st); prng = rng.clone()
    tr = gogame.batch_track(st); trng = rng.clone()
    idx = np.arange(0, B, every); idx_t = torch.as_tensor(idx, device='cuda')
    want = np.zeros((len(idx), 6, N, N), np.uint8)
    orng = np.array([c_oracle.lib().gg_oracle_rng_seed(seed, int(i)) for i in idx], dtype=np.uint64)
    obs = torch.empty_like(st)
    gen = np.random.default_rng(seed)
    t = 0
    while t < plies:
        k = int(gen.integers(2, 70))
        gogame.batch_rollout(st, rng, k, True)
        gogame.batch_rollout_packed(pk, prng, k, True)
        mode = int(gen.integers(0, 3))
        if mode == 0:
            gogame.batch_rollout_tracked(tr, trng, k, True)
        elif mode == 1:
            for _ in range(k):
                gogame.batch_env_step_tracked(tr, None, trng, 6.5, 'real', True, states_out=obs)
        else:
            for _ in range(k):
                gogame.batch_rollout_tracked(tr, trng, 1, True)
        want, orng, _ = c_oracle.batch_rollout_mt(want, orng, k, True)
        t += k
        assert np.array_equal(st[idx_t].cpu().numpy(), want), (N, B, t)
        assert torch.equal(gogame.batch_unpack(pk, N), st) and torch.equal(prng, rng), (N, B, t, 'packed')
        assert torch.equal(gogame.batch_untrack(tr), st) and torch.equal(trng, rng), (N, B, t, 'tracked', mode)
        if mode == 1:
            assert torch.equal(obs, st), (N, B, t, 'observation')
        if t % 5 == 0:
            assert torch.equal(tr, gogame.batch_track(st)), (N, B, t, 'classes')
    assert torch.equal(tr, gogame.batch_track(st)), (N, B, t, 'classes')
    print('multi-ply soak %dx%d: %d games x %d plies bit-exact (bytes / packed / tracked / env step; %d games vs oracle)'
          % (N, N, B, t, len(idx)), flush=True)


def kids(N, B, seed):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, seed)
    per = max(1, B // 8)
    for g in range(8):
        gogame.batch_rollout(st[g * per:(g + 1) * per], rng[g * per:(g + 1) * per], 10 + (N * N * g) // 7, False)
    st = st[st[:, 5, 0, 0] == 0].contiguous()
    host = st.cpu().numpy()
    for canon in (False, True):
        got = gogame.batch_children(st, canonical=canon).cpu().numpy()
        assert np.array_equal(got, c_oracle.batch_children(host, canon)), (N, canon)
    print('children %dx%d: %d parents of every phase bit-exact' % (N, N, len(st)), flush=True)

t0 = time.time()
if len(sys.argv) > 1 and sys.argv[1] == 'multi':
    soak_multi_ply(19, 16384, 2600, 11); soak_multi_ply(19, 8200, 1500, 12); soak_multi_ply(13, 12288, 1500, 13)
    soak_multi_ply(9, 16400, 1200, 14); soak_multi_ply(5, 8192, 500, 15); soak_multi_ply(2, 9000, 100, 16)
    print('done in %.0f s' % (time.time() - t0)); sys.exit(0)
soak(19, 192, 2400, 1); soak(13, 256, 1500, 2); soak(9, 384, 1200, 3); soak(6, 256, 600, 4); soak(3, 128, 300, 5)
kids(19, 384, 6); kids(13, 256, 7); kids(9, 512, 8); kids(4, 128, 9)
print('done in %.0f s' % (time.time() - t0))
