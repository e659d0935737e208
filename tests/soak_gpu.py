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
            k += 8    # long enough for the v3 kernel (used from 8 192 games up)
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
soak(19, 192, 2400, 1); soak(13, 256, 1500, 2); soak(9, 384, 1200, 3); soak(6, 256, 600, 4); soak(3, 128, 300, 5)
kids(19, 384, 6); kids(13, 256, 7); kids(9, 512, 8); kids(4, 128, 9)
print('done in %.0f s' % (time.time() - t0))
