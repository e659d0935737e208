"""-m gpu: the thirty-two-board multi-ply kernel (gymgo_amd/csrc/gg_v5.h, k_rollout5: a pair of lanes per board, the floods of a
ply as a compacted job list) against the pinned C oracle - states, generator states, last actions, step counters.

gg_batch_rollout / gg_batch_rollout_tracked hand full-size 19x19 launches of >= 8 plies to it above 128 games per CU
(gg_kernels.hip: use_rollout5; 9x9 and 13x13: from 160 games per CU), i.e. from 32 769 games on the whole device: the BASELINE config-3 test (test_gpu_configs.py) and the
bench's own driver (test_gpu_deep.py) run it at that size.  Here the library is sized for FOUR compute units
(GYMGO_AMD_CUS=4, read once per process: a process of its own), so that 513 games take the kernel and the oracle can replay
whole games: both sides of the games / plies take-over, ragged last waves (a wave of 2 .. 32 boards), frozen games, resets, and a
crafted position whose first ply posts MORE flood jobs than a wave has lanes (a second job batch).
Reference loop: gym_go/envs/go_env.py:49-81 over gym_go/gogame.py:34-87.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
import torch
from gymgo_amd import gogame, _lib
from oracle import c_oracle
N = 19
assert _lib.lib().gg_device_cus() == 4


def run(states, B, launches, auto_reset, seed, tracked):
    st = torch.from_numpy(states).cuda()
    rng = gogame.rng_seed(B, seed, 0, 'cuda')
    want, want_rng = states.copy(), rng.cpu().numpy().view(np.uint64).copy()
    sd = torch.zeros(B, dtype=torch.int64, device='cuda')
    tr = gogame.batch_track(st) if tracked else None
    played = 0
    for F in launches:
        la = torch.full((B,), -9, dtype=torch.int32, device='cuda')
        if tracked:
            gogame.batch_rollout_tracked(tr, rng, F, auto_reset, la, sd)
        else:
            gogame.batch_rollout(st, rng, F, auto_reset, la, sd)
        want, want_rng, want_last = c_oracle.batch_rollout_mt(want, want_rng, F, auto_reset)
        got = gogame.batch_untrack(tr).cpu().numpy() if tracked else st.cpu().numpy()
        bad = np.flatnonzero((got != want).reshape(B, -1).any(axis=1))
        assert len(bad) == 0, (B, F, auto_reset, tracked, bad[:6].tolist())
        assert np.array_equal(rng.cpu().numpy().view(np.uint64), want_rng), (B, F, tracked)
        assert np.array_equal(la.cpu().numpy(), want_last), (B, F, tracked)
        played += F
    if auto_reset:
        assert np.array_equal(sd.cpu().numpy(), np.full(B, played, np.int64))
    else:
        assert int(sd.max()) <= played and int(sd.min()) > 0
''' % ROOT

SMALL = PRELUDE + r'''
# 9x9 / 13x13 on 4 CUs: the kernel takes launches of >= 8 plies from 640 games on (160 per CU); below that and for shorter launches
# the one-row-per-lane and sixteen-board kernels serve the call
for N in (9, 13):
    for B in (639, 640, 641, 700, 1025, 2049):
        empty = np.zeros((B, 6, N, N), np.uint8)
        for tracked in (False, True):
            run(empty, B, (7, 8, 9, 60, 200), True, 300 + B + N, tracked)
    run(np.zeros((1100, 6, N, N), np.uint8), 1100, (3 * N * N + 9, 8, 150), True, 17 + N, False)      # whole games: passes, resets
    run(np.zeros((1100, 6, N, N), np.uint8), 1100, (3 * N * N + 9, 8, 300), False, 18 + N, False)     # frozen games
    run(np.zeros((1030, 6, N, N), np.uint8), 1030, (2 * N * N, 400, 9), False, 19 + N, True)
print('R5 OK')
'''

TAKE_OVER = PRELUDE + r'''
# 4 CUs: the kernel takes launches of >= 8 plies above 512 games; 512 games / 7 plies stay with the other families
for B in (512, 513, 514, 600, 1025, 2049):
    empty = np.zeros((B, 6, N, N), np.uint8)
    for tracked in (False, True):
        run(empty, B, (7, 8, 9, 90, 300), True, 100 + B, tracked)
run(np.zeros((1100, 6, N, N), np.uint8), 1100, (3 * N * N + 9, 8, 200), True, 7, False)      # whole games: passes, resets
run(np.zeros((1100, 6, N, N), np.uint8), 1100, (3 * N * N + 9, 8, 400), False, 8, False)     # frozen games neither move nor draw
run(np.zeros((1030, 6, N, N), np.uint8), 1030, (2 * N * N, 700, 9), False, 9, True)
print('R5 OK')
'''

JOBS = PRELUDE + r'''
# White stones on every second point except a sparse grid of holes, black to move: an empty point next to a hole touches THREE
# separate one-stone white groups (three flood jobs), every other empty point between four white stones is suicide, the holes are
# playable too (no job).  Four of five legal moves post three jobs: ~77 jobs per wave of 32 boards on the first ply, more than
# its 64 lanes - the kernel's second job batch.  The boards differ only in their generators.
s0 = np.zeros((6, N, N), np.uint8)
for r in range(N):
    for c in range(N):
        if (r + c) % 2 == 0 and not (r % 4 == 2 and c % 4 == 2):
            s0[1, r, c] = 1
s0[3] = c_oracle.compute_invalid_moves(s0, 1)     # (the mask of the player who moves AFTER player 1: black)
legal = (s0[3] == 0) & (s0[1] == 0)
wn = np.zeros((N, N), np.int64)           # white neighbours of every point = the flood jobs a black stone there posts
wn[1:] += s0[1, :-1]; wn[:-1] += s0[1, 1:]; wn[:, 1:] += s0[1, :, :-1]; wn[:, :-1] += s0[1, :, 1:]
assert int((legal & (wn == 3)).sum()) >= 60 and int((legal & (wn == 0)).sum()) >= 12, (int(legal.sum()),)
B = 1056
states = np.repeat(s0[None], B, axis=0)
_, _, last1 = c_oracle.batch_rollout_mt(states.copy(), gogame.rng_seed(B, 42, 0, 'cuda').cpu().numpy().view(np.uint64).copy(), 1, True)
jobs = np.where(last1 < N * N, wn.reshape(-1)[np.minimum(last1, N * N - 1)], 0)
per_wave = jobs.reshape(-1, 32).sum(axis=1)
assert int(per_wave.max()) > 64 and int((per_wave > 64).sum()) >= 8, per_wave.tolist()
for tracked in (False, True):
    run(states, B, (8, 1, 30), True, 42, tracked)
print('R5 OK')
'''


def _run(script):
    env = dict(os.environ)
    env['GYMGO_AMD_CUS'] = '4'
    p = subprocess.run([sys.executable, '-c', script], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert 'R5 OK' in p.stdout


def test_r5_both_sides_of_its_take_over_ragged_waves_frozen_games():
    _run(TAKE_OVER)


def test_r5_small_boards_both_sides_of_their_take_over():
    _run(SMALL)


def test_r5_second_job_batch_on_a_crafted_position():
    _run(JOBS)


@pytest.mark.parametrize('size', [19, 13, 9])
def test_r5_full_device_batch_same_as_chunks_on_the_other_kernels(size):
    """65 536 games on the whole device (the kernel's own shape, two waves per SIMD) against the same games run in chunks small
    enough to stay on the sixteen-board / one-row-per-lane kernels, which test_gpu_lat.py and test_gpu_configs.py hold to the oracle."""
    from gymgo_amd import gogame, _lib
    cus = int(_lib.lib().gg_device_cus())
    B = cus * 256 + 37   # (two waves of 32 boards per SIMD and a ragged last wave)
    for tracked in (False, True):
        st = gogame.batch_init_state(B, size, device='cuda')
        rng = gogame.rng_seed(B, 99, 0, 'cuda')
        ch = B // 16
        for g in range(1, 16):      # de-synchronised games
            gogame.batch_rollout(st[g * ch:(g + 1) * ch], rng[g * ch:(g + 1) * ch], 37 * g, True)
        st2, rng2 = st.clone(), rng.clone()
        la, la2 = (torch.full((B,), -3, dtype=torch.int32, device='cuda') for _ in range(2))
        sd, sd2 = (torch.zeros(B, dtype=torch.int64, device='cuda') for _ in range(2))
        part = cus * 64
        if tracked:
            tr, tr2 = gogame.batch_track(st), gogame.batch_track(st2)
        for F in (8, 61, 256):
            if tracked:
                gogame.batch_rollout_tracked(tr, rng, F, True, la, sd)
                for lo in range(0, B, part):
                    gogame.batch_rollout_tracked(tr2[lo:lo + part], rng2[lo:lo + part], F, True, la2[lo:lo + part], sd2[lo:lo + part])
                assert torch.equal(tr, tr2), F
            else:
                gogame.batch_rollout(st, rng, F, True, la, sd)
                for lo in range(0, B, part):
                    gogame.batch_rollout(st2[lo:lo + part], rng2[lo:lo + part], F, True, la2[lo:lo + part], sd2[lo:lo + part])
                assert torch.equal(st, st2), F
            assert torch.equal(rng, rng2) and torch.equal(la, la2) and torch.equal(sd, sd2), (tracked, F)
