"""ctypes front-end of oracle/libgg_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg - as the
checker, never by the product path (gymgo_amd/).  All arrays are NumPy uint8 / int32 on the host.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libgg_oracle.so')
_lib = None

_u8p = ctypes.POINTER(ctypes.c_uint8)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u64p = ctypes.POINTER(ctypes.c_uint64)


def build(force=False):
    src = os.path.join(_HERE, 'gg_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B'])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.gg_oracle_next_state.restype = ctypes.c_int32
        L.gg_oracle_next_state.argtypes = [_u8p, ctypes.c_int32, _u8p, ctypes.c_int32, ctypes.c_int32]
        L.gg_oracle_batch_next_states.restype = None
        L.gg_oracle_batch_next_states.argtypes = [_u8p, _i32p, _u8p, _i32p, ctypes.c_int64, ctypes.c_int32,
                                                  ctypes.c_int32]
        L.gg_oracle_compute_invalid_moves.restype = None
        L.gg_oracle_compute_invalid_moves.argtypes = [_u8p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _u8p]
        L.gg_oracle_batch_areas.restype = None
        L.gg_oracle_batch_areas.argtypes = [_u8p, _i32p, _i32p, ctypes.c_int64, ctypes.c_int32]
        L.gg_oracle_batch_children.restype = None
        L.gg_oracle_batch_children.argtypes = [_u8p, _u8p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]
        L.gg_oracle_canonical_form.restype = None
        L.gg_oracle_canonical_form.argtypes = [_u8p, ctypes.c_int32]
        L.gg_oracle_rng_seed.restype = ctypes.c_uint64
        L.gg_oracle_rng_seed.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
        L.gg_oracle_batch_rollout.restype = None
        L.gg_oracle_batch_rollout.argtypes = [_u8p, _u64p, _i32p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                              ctypes.c_int32]
        L.gg_oracle_update_pieces.restype = ctypes.c_int32
        L.gg_oracle_update_pieces.argtypes = [_u8p, ctypes.c_int32, _i32p, ctypes.c_int32, ctypes.c_int32, _u8p]
        L.gg_oracle_batch_sample_weighted.restype = None
        L.gg_oracle_batch_sample_weighted.argtypes = [_u8p, ctypes.POINTER(ctypes.c_float), _u64p, _i32p, ctypes.c_int64,
                                                      ctypes.c_int32]
        _lib = L
    return _lib


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(_u8p)


def batch_next_states(states, actions, canonical=False):
    """states [B,6,N,N] uint8, actions [B] -> (next [B,6,N,N] uint8, status [B] int32)."""
    states, sp = _u8(states)
    B, _, N, _ = states.shape
    actions = np.ascontiguousarray(actions, dtype=np.int32)
    out = np.empty_like(states)
    status = np.zeros(B, dtype=np.int32)
    lib().gg_oracle_batch_next_states(sp, actions.ctypes.data_as(_i32p), out.ctypes.data_as(_u8p),
                                      status.ctypes.data_as(_i32p), B, N, int(bool(canonical)))
    return out, status


def next_state(state, action, canonical=False):
    out, status = batch_next_states(np.asarray(state)[None], np.array([action]), canonical)
    if status[0]:
        raise AssertionError(('Invalid move', int(action)))
    return out[0]


def compute_invalid_moves(state, player, ko=-1):
    state, sp = _u8(state)
    N = state.shape[-1]
    mask = np.empty((N, N), dtype=np.uint8)
    lib().gg_oracle_compute_invalid_moves(sp, N, int(player), int(ko), mask.ctypes.data_as(_u8p))
    return mask


def batch_areas(states):
    states, sp = _u8(states)
    B, _, N, _ = states.shape
    black = np.empty(B, dtype=np.int32)
    white = np.empty(B, dtype=np.int32)
    lib().gg_oracle_batch_areas(sp, black.ctypes.data_as(_i32p), white.ctypes.data_as(_i32p), B, N)
    return black, white


def batch_children(states, canonical=False):
    states, sp = _u8(states)
    B, _, N, _ = states.shape
    out = np.empty((B, N * N + 1, 6, N, N), dtype=np.uint8)
    lib().gg_oracle_batch_children(sp, out.ctypes.data_as(_u8p), B, N, int(bool(canonical)))
    return out


def canonical_form(state):
    state = np.array(state, dtype=np.uint8, copy=True, order='C')
    lib().gg_oracle_canonical_form(state.ctypes.data_as(_u8p), state.shape[-1])
    return state


def rng_seed(base_seed, B):
    L = lib()
    return np.array([L.gg_oracle_rng_seed(int(base_seed), b) for b in range(B)], dtype=np.uint64)


def batch_rollout(states, rng, plies, auto_reset=True):
    """In place on copies; returns (states, rng, last_actions)."""
    states = np.array(states, dtype=np.uint8, copy=True, order='C')
    rng = np.array(rng, dtype=np.uint64, copy=True)
    B, _, N, _ = states.shape
    last = np.full(B, -1, dtype=np.int32)
    lib().gg_oracle_batch_rollout(states.ctypes.data_as(_u8p), rng.ctypes.data_as(_u64p),
                                  last.ctypes.data_as(_i32p), B, N, int(plies), int(bool(auto_reset)))
    return states, rng, last


def batch_sample_weighted(states, weights, rng):
    """Policy-weighted draw per game (see gg_oracle_sample_weighted): states [B,6,N,N] (plane 3 masks) or None,
    weights float32 [B,N*N+1], rng uint64 [B] -> (actions int32 [B], rng after)."""
    weights = np.ascontiguousarray(weights, dtype=np.float32)
    B = weights.shape[0]
    N = int(round((weights.shape[1] - 1) ** 0.5))
    assert N * N + 1 == weights.shape[1]
    rng = np.array(rng, dtype=np.uint64, copy=True)
    actions = np.empty(B, dtype=np.int32)
    if states is None:
        sp = None
    else:
        states, sp = _u8(states)
        assert states.shape == (B, 6, N, N)
    lib().gg_oracle_batch_sample_weighted(sp, weights.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                          rng.ctypes.data_as(_u64p), actions.ctypes.data_as(_i32p), B, N)
    return actions, rng


def update_pieces(state, adj_flat, player):
    """state_utils.update_pieces on a copy -> (state after, killed mask [N,N], number of killed groups)."""
    state = np.array(state, dtype=np.uint8, copy=True, order='C')
    N = state.shape[-1]
    adj = np.ascontiguousarray(adj_flat, dtype=np.int32).reshape(-1)
    killed = np.zeros((N, N), dtype=np.uint8)
    n = lib().gg_oracle_update_pieces(state.ctypes.data_as(_u8p), N, adj.ctypes.data_as(_i32p), len(adj), int(player),
                                      killed.ctypes.data_as(_u8p))
    return state, killed, n


def _chunks(B, workers):
    step = max(1, (B + workers - 1) // workers)
    return [(lo, min(B, lo + step)) for lo in range(0, B, step)]


def parallel(fn, B, workers=None):
    """Run fn(lo, hi) over slices of a batch on a thread pool (the C calls release the GIL): the oracle stays a
    one-thread-per-board restatement, large parity batches just use every host core."""
    import concurrent.futures as cf
    workers = workers or min(32, os.cpu_count() or 1)
    parts = _chunks(B, workers)
    if len(parts) <= 1:
        return [fn(0, B)]
    with cf.ThreadPoolExecutor(len(parts)) as ex:
        return list(ex.map(lambda p: fn(*p), parts))


def batch_rollout_mt(states, rng, plies, auto_reset=True, workers=None):
    """batch_rollout over host threads -> (states, rng, last_actions)."""
    B = len(states)
    out = parallel(lambda lo, hi: batch_rollout(states[lo:hi], rng[lo:hi], plies, auto_reset), B, workers)
    return (np.concatenate([o[0] for o in out]), np.concatenate([o[1] for o in out]), np.concatenate([o[2] for o in out]))


def batch_next_states_mt(states, actions, canonical=False, workers=None):
    B = len(states)
    out = parallel(lambda lo, hi: batch_next_states(states[lo:hi], actions[lo:hi], canonical), B, workers)
    return np.concatenate([o[0] for o in out]), np.concatenate([o[1] for o in out])


def batch_children_mt(states, canonical=False, workers=None):
    B = len(states)
    return np.concatenate(parallel(lambda lo, hi: batch_children(states[lo:hi], canonical), B, workers))


def batch_areas_mt(states, workers=None):
    B = len(states)
    out = parallel(lambda lo, hi: batch_areas(states[lo:hi]), B, workers)
    return np.concatenate([o[0] for o in out]), np.concatenate([o[1] for o in out])
