#!/usr/bin/env python
"""Time-boxed differential fuzz of the HIP path against the oracle (run on the GPU box; not collected by pytest):

    python tests/fuzz_parity.py [seconds=120] [seed=1]

Every round draws a board size (2 .. 19), a batch, a layout, a launch length and a mix of game phases, then walks the same
games on the device (fused rollouts in one to three launches, one env step with rewards, one policy-weighted step with
float32 / bfloat16 / float16 weights, next_states with drawn - partly illegal - moves, children of a few parents (padded and un-padded), areas, the invalid-move mask, track_states followed by a fused rollout on the tracked boards)
and through oracle/gg_oracle.c, and compares
boards, generators, moves, rewards and masks bit for bit.  The seeded tests of the suite pin known cases; this looks for
the cases nobody wrote down (rare capture / ko / suicide sub-paths of the multi-ply kernel take thousands of plies to hit).
Exit code 1 on the first mismatch, with the round's parameters."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gymgo_amd import gogame                      # noqa: E402
from gymgo_amd.envs import GoVecEnv               # noqa: E402
from oracle import c_oracle                       # noqa: E402  (the checker)


def fail(what, params):
    print('MISMATCH: %s\n  round parameters: %r' % (what, params), flush=True)
    sys.exit(1)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rs = np.random.default_rng(seed)
    t_end = time.time() + budget
    rounds = steps = 0
    while time.time() < t_end:
        N = int(rs.choice([2, 3, 4, 5, 6, 7, 8, 9, 9, 10, 11, 12, 13, 13, 14, 15, 16, 17, 18, 19, 19, 19]))
        B = int(rs.choice([1, 3, 17, 64, 257, 1000, 2048, 3000]))
        layout = str(rs.choice(['bytes', 'packed', 'tracked']))
        auto = bool(rs.integers(0, 2))
        base = int(rs.integers(1, 1 << 30))
        params = dict(N=N, B=B, layout=layout, auto_reset=auto, seed=base)
        env = GoVecEnv(B, N, komi=float(rs.choice([0.0, 0.5, 7.5])), reward_method=str(rs.choice(['real', 'heuristic'])),
                       seed=base, layout=layout, auto_reset=auto)
        st = np.zeros((B, 6, N, N), dtype=np.uint8)
        rng = env.rng.cpu().numpy().view(np.uint64).copy()
        # fused rollouts: one to three launches of random lengths (1 .. ~3 N^2 plies in total)
        for _ in range(int(rs.integers(1, 4))):
            F = int(rs.integers(1, max(2, N * N)))
            params['plies'] = params.get('plies', []) + [F]
            env.rollout(F)
            st, rng, last = c_oracle.batch_rollout_mt(st, rng, F, auto)
            steps += B * F
        got = env.states.cpu().numpy()
        if not np.array_equal(got, st):
            fail('boards after the fused rollouts (%d games differ)' % int((got != st).any(axis=(1, 2, 3)).sum()), params)
        if not np.array_equal(env.rng.cpu().numpy().view(np.uint64), rng):
            fail('generators after the fused rollouts', params)
        # one env step with device-drawn moves == one oracle ply (+ areas for the reward)
        obs, rewards, dones, status = env.step()
        st2, rng2, last2 = c_oracle.batch_rollout_mt(st, rng, 1, auto)
        frozen = (st[:, 5, 0, 0] == 1) & (not auto)
        if not np.array_equal(env.states.cpu().numpy()[~frozen], st2[~frozen]):   # (layout 'packed' returns its packed rows)
            fail('observation of GoVecEnv.step', params)
        if not np.array_equal(env.last_actions.cpu().numpy()[~frozen], last2[~frozen]):
            fail('moves drawn by GoVecEnv.step', params)
        b, w = c_oracle.batch_areas_mt(st2)
        margin = b.astype(np.float64) - w - env.komi
        over = st2[:, 5, 0, 0] == 1
        if env.reward_method == 'real':
            want = np.where(over, np.sign(margin), 0.0)
        else:
            want = np.where(over, np.where(margin > 0, 1.0, -1.0) * N * N, margin)
        if not np.array_equal(rewards.cpu().numpy().astype(np.float64)[~frozen], want[~frozen]):
            fail('rewards of GoVecEnv.step', params)
        # one policy-weighted step (float32 / bfloat16 / float16 weights, some rows all zero): reset, draw, refuse, step
        wdt = [torch.float32, torch.bfloat16, torch.float16][int(rs.integers(0, 3))]
        w = (rs.random((B, N * N + 1)) ** 3).astype(np.float32)
        w[rs.random(B) < 0.02] = 0.0
        wd = torch.from_numpy(w).cuda().to(wdt)
        params['weights'] = str(wdt)
        start = st2.copy()
        over0 = start[:, 5, 0, 0] == 1
        frozen2 = over0 & (not auto)
        if auto:
            start[over0] = 0
        acts_w = np.full(B, -1, np.int32)
        rng3 = rng2.copy()
        if (~frozen2).any():
            a, r = c_oracle.batch_sample_weighted(start[~frozen2], wd.float().cpu().numpy()[~frozen2], rng2[~frozen2])
            acts_w[~frozen2], rng3[~frozen2] = a, r
        okw = np.flatnonzero(acts_w >= 0)
        want_w = start.copy()
        if len(okw):
            want_w[okw] = c_oracle.batch_next_states(start[okw], acts_w[okw])[0]
        o3, r3, d3, s3 = env.step(probs=wd)
        if not np.array_equal(env.last_actions.cpu().numpy(), acts_w):
            fail('moves drawn from policy weights', params)
        if not (np.array_equal(env.states.cpu().numpy(), want_w) and np.array_equal(env.rng.cpu().numpy().view(np.uint64), rng3)):
            fail('boards / generators after the policy-weighted step', params)
        if not np.array_equal(s3.cpu().numpy(), (acts_w < 0).astype(np.int32)):
            fail('status of the policy-weighted step', params)
        st2 = want_w
        b, w_area = c_oracle.batch_areas_mt(st2)
        w = w_area
        # next_states with moves drawn uniformly from ALL actions (many illegal), children, areas, mask on these positions
        dev = torch.from_numpy(st2).cuda()
        acts = rs.integers(0, N * N + 1, size=B).astype(np.int32)
        nxt, stat = gogame.batch_next_states(dev, torch.from_numpy(acts).cuda(), check=False)
        wn, ws = c_oracle.batch_next_states_mt(st2, acts)
        playable = st2[:, 5, 0, 0] == 0   # (the reference plays on after the end; the oracle follows it, so do we)
        if not (np.array_equal(nxt.cpu().numpy(), wn) and np.array_equal(stat.cpu().numpy(), ws)):
            fail('batch_next_states with arbitrary moves (%d playable)' % int(playable.sum()), params)
        k = min(B, 24)
        kids_want = c_oracle.batch_children_mt(st2[:k])
        if not np.array_equal(gogame.batch_children(dev[:k]).cpu().numpy(), kids_want):
            fail('batch_children', params)
        # the un-padded form: the slots valid_moves() keeps (plane 3 clear + the pass; every action once the game has ended)
        keep = np.concatenate([st2[:k, 3].reshape(k, -1) == 0, np.ones((k, 1), bool)], axis=1)
        keep[st2[:k, 5, 0, 0] == 1] = True
        ckids, coffs = gogame.batch_children(dev[:k], padded=False)
        if not (np.array_equal(coffs.cpu().numpy(), np.concatenate([[0], np.cumsum(keep.sum(axis=1))]).astype(np.int32))
                and np.array_equal(ckids.cpu().numpy(), kids_want[keep])):
            fail('batch_children(padded=False)', params)
        gb, gw = gogame.batch_areas(dev)
        if not (np.array_equal(gb.cpu().numpy(), b) and np.array_equal(gw.cpu().numpy(), w)):
            fail('batch_areas', params)
        # (plane 3 carries the ko point of the last move, which planes 0 - 2 cannot tell: the recomputed mask may lack
        # exactly that one point and nothing else)
        mask = gogame._invalid_mask_dev(dev).cpu().numpy()
        extra = (mask[playable] == 1) & (st2[playable, 3] == 0)
        missing = (mask[playable] == 0) & (st2[playable, 3] == 1)
        if extra.any() or (missing.sum(axis=(1, 2)) > 1).any():
            fail('invalid-move mask recomputed from planes 0-2', params)
        # byte planes -> tracked boards (from-scratch liberty classes) -> a fused rollout that lives on those classes
        tr = gogame.batch_track(dev)
        if not torch.equal(gogame.batch_untrack(tr), dev):
            fail('untrack(track(states))', params)
        F2 = int(rs.integers(1, 2 * N + 2))
        r_dev = torch.from_numpy(rng3.view(np.int64)).cuda()
        gogame.batch_rollout_tracked(tr, r_dev, F2, auto)
        st4, rng4, _ = c_oracle.batch_rollout_mt(st2, rng3, F2, auto)
        if not (np.array_equal(gogame.batch_untrack(tr).cpu().numpy(), st4) and np.array_equal(r_dev.cpu().numpy().view(np.uint64), rng4)):
            fail('fused rollout on freshly tracked boards (%d plies)' % F2, params)
        steps += B * F2
        rounds += 1
    print('fuzz ok: %d rounds, %.2e oracle-checked plies in %.0f s (seed %d)' % (rounds, steps, budget, seed), flush=True)


if __name__ == '__main__':
    main()
