"""-m gpu: the C-ABI from several host threads at once (include/gymgo_amd.h: "every entry point is re-entrant and
thread-safe").  Two threads, each with its own HIP stream and its own games, drive the entry points whose host side
touches the library's mutable state - the per-device CU cache, the mutex-guarded occupancy cache behind the age split
(gg_batch_invalid_mask, gg_batch_track_states, gg_batch_env_step ...), the FairShare board of the fused launches - in
lock-step from a barrier, so that a cold process hits the first-use paths concurrently.  Results must be what the same
calls give one after the other.  tools/sanitize.sh runs this file (alone, in a fresh process) against an
-fsanitize=address,undefined build of the host side.  The reference has no counterpart (pure Python, module constants
only: gym_go/state_utils.py:7-21)."""
import hashlib
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def _workload(seed, size, games, rounds, stream=None, barrier=None):
    """A mix of per-ply, fused and env-step calls on one private batch; returns a digest of everything it produced."""
    from gymgo_amd import gogame, state_utils
    dev = torch.device('cuda', 0)
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream(dev))
    h = hashlib.sha256()
    with ctx:
        st = gogame.batch_init_state(games, size, device=dev)
        rng = gogame.rng_seed(games, seed, 0, dev)
        if barrier is not None:
            barrier.wait()
        for r in range(rounds):
            gogame.batch_rollout(st, rng, 24, True)                         # fused launch: FairShare board
            mask = state_utils.batch_compute_invalid_moves(st, None, None)   # gg_batch_invalid_mask: launch_pairs -> occupancy cache
            acts = gogame.batch_sample_actions(st, rng)
            nxt, status = gogame.batch_next_states(st, acts, check=False)   # per-ply kernels: occupancy cache, age split
            tracked = gogame.batch_track(nxt)
            rewards, dones, stat2, taken = gogame.batch_env_step_tracked(tracked, None, rng, 7.5, 'real', True)
            st = gogame.batch_untrack(tracked)
            areas = gogame.batch_areas(st)
            for t in (nxt, status, rewards, dones, taken, st, rng, mask) + tuple(areas):
                h.update(np.ascontiguousarray(t.cpu().numpy()).tobytes())
        (stream or torch.cuda.current_stream(dev)).synchronize()
    return h.hexdigest()


@pytest.mark.parametrize('size,games', [(19, 40000), (9, 5000)])
def test_two_threads_two_streams_give_the_results_of_sequential_calls(size, games):
    want = [_workload(101, size, games, 3), _workload(202, size, games + 37, 3)]
    got, errors = [None, None], []
    barrier = threading.Barrier(2)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def run(i, seed, n):
        try:
            torch.cuda.set_device(0)
            got[i] = _workload(seed, size, n, 3, streams[i], barrier)
        except Exception as e:          # surfaced in the main thread
            errors.append(repr(e))
            try:
                barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=run, args=(0, 101, games)), threading.Thread(target=run, args=(1, 202, games + 37))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert got == want
