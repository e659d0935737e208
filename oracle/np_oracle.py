"""NumPy/SciPy restatement of the reference's per-game step.  TEST INFRASTRUCTURE ONLY.

Used (a) as a second, independently written oracle and (b) as bench.py's `cpu_baseline`
(kind "port"): it executes the same SciPy primitives per step as the reference does
(`ndimage.label` x3, one `binary_dilation` per neighbouring opponent group, two stacked
`binary_dilation` over [n_groups, N, N], one `convolve`), so its speed on the GPU box's host
cores stands in for the reference, which cannot travel there.  Pinned against the real
reference by oracle/ref_harness/pin_oracle.py (bit-exact; speed ratio recorded in DESIGN.md).

Never imported by the product path (gymgo_amd/).
States are float64 or uint8 arrays [6, N, N] with values 0/1 (gym_go/gogame.py:7-19).
"""
import numpy as np
from scipy import ndimage

BLACK, WHITE, TURN, INVD, PASSED, DONE = range(6)   # gym_go/govars.py:4-9
CROSS_NO_CENTRE = np.array([[0, 1, 0], [1, 0, 1], [0, 1, 0]])  # gym_go/state_utils.py:17-19
_STEPS = ((-1, 0), (1, 0), (0, -1), (0, 1))                   # gym_go/state_utils.py:21


def _liberty_stack(stones, vacant):
    """Per-group liberty maps [n_groups, N, N] and their sizes (gym_go/state_utils.py:48-65)."""
    labels, n = ndimage.label(stones)
    stack = np.zeros((n,) + stones.shape)
    for g in range(n):
        stack[g] = labels == g + 1
    libs = vacant[None] * ndimage.binary_dilation(stack, CROSS_NO_CENTRE[None])
    return libs, libs.sum(axis=(1, 2))


def invalid_mask(state, mover, ko=None):
    """gym_go/state_utils.py:24-83: mask of points the side AFTER `mover` may not play."""
    occupied = state[BLACK] + state[WHITE]
    vacant = 1 - occupied
    mine, n_mine = _liberty_stack(state[mover], vacant)
    theirs, n_theirs = _liberty_stack(state[1 - mover], vacant)
    maybe_bad = mine[n_mine > 1].sum(axis=0) + theirs[n_theirs == 1].sum(axis=0)      # :70-71
    surely_ok = mine[n_mine == 1].sum(axis=0) + theirs[n_theirs > 1].sum(axis=0)      # :73-74
    boxed_in = ndimage.convolve(occupied, CROSS_NO_CENTRE, mode='constant', cval=1) == 4   # :77
    bad = occupied + maybe_bad * (surely_ok == 0) * boxed_in                           # :78
    if ko is not None:
        bad[ko[0], ko[1]] = 1                                                          # :81-82
    return bad > 0


def _capture(state, around, mover):
    """gym_go/state_utils.py:159-180: drop opponent groups next to the new stone with no liberty."""
    foe = 1 - mover
    vacant = 1 - (state[BLACK] + state[WHITE])
    labels, _ = ndimage.label(state[foe])
    dead = []
    for g in np.unique(labels[around[:, 0], around[:, 1]]):
        if g == 0:
            continue
        body = labels == g
        if (vacant * ndimage.binary_dilation(body)).sum() <= 0:
            where = np.argwhere(body)
            state[foe, where[:, 0], where[:, 1]] = 0
            dead.append(where)
    return dead


def next_state(state, action, canonical=False):
    """gym_go/gogame.py:34-87."""
    s = np.array(state, copy=True)
    n = s.shape[1]
    mover = int(s[TURN].max())                       # :44, :241-246
    ko = None
    if action == n * n:                              # :48-53
        if (s[PASSED] == 1).max():
            s[DONE] = 1
        s[PASSED] = 1
    else:
        r, c = action // n, action % n
        s[PASSED] = 0
        assert s[INVD, r, c] == 0, ('Invalid move', (r, c))                  # :59
        s[mover, r, c] = 1
        around = np.array([(r + dr, c + dc) for dr, dc in _STEPS
                           if 0 <= r + dr < n and 0 <= c + dc < n])         # state_utils.py:214-223
        boxed = bool((s[1 - mover][around[:, 0], around[:, 1]] > 0).all())
        dead = _capture(s, around, mover)
        if len(dead) == 1 and boxed and len(dead[0]) == 1:                   # :72-75
            ko = dead[0][0]
    s[INVD] = invalid_mask(s, mover, ko)             # :78
    s[TURN] = 1 - s[TURN]                            # :81
    if canonical and s[TURN].max() == 1:             # :83-85, :313-321
        s[[BLACK, WHITE]] = s[[WHITE, BLACK]]
        s[TURN] = 0
    return s


def areas(state):
    """gym_go/gogame.py:275-300."""
    vacant = 1 - (state[BLACK] + state[WHITE])
    labels, n = ndimage.label(vacant)
    b, w = state[BLACK].sum(), state[WHITE].sum()
    for g in range(1, n + 1):
        region = labels == g
        rim = ndimage.binary_dilation(region)
        touches_b = (state[BLACK] * rim > 0).any()
        touches_w = (state[WHITE] * rim > 0).any()
        if touches_b and not touches_w:
            b += region.sum()
        elif touches_w and not touches_b:
            w += region.sum()
    return b, w


def random_rollout_steps(size, seconds, seed):
    """Uniform-random self-play (GoEnv.uniform_random_action, gym_go/envs/go_env.py:78-81) with
    auto-reset for `seconds` of wall time; returns the number of steps done.  cpu_baseline worker."""
    import time
    rng = np.random.default_rng(seed)
    s = np.zeros((6, size, size))
    steps = 0
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        if s[DONE].max() == 1:
            s = np.zeros((6, size, size))
        ok = np.flatnonzero(np.append(s[INVD].ravel(), 0) == 0)
        s = next_state(s, int(rng.choice(ok)))
        steps += 1
    return steps
