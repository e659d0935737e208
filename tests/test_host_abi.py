"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/gymgo_amd.h declares;
host-side logic (sharding, env plumbing that needs no device).  No compute calls without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built(native_built):
    # (conftest.native_built builds the library when it is missing or stale and skips on a box without hipcc / torch)
    from gymgo_amd import _lib
    return _lib


def test_header_symbols_exported(built):
    hdr = open(os.path.join(ROOT, 'include', 'gymgo_amd.h')).read()
    declared = set(re.findall(r'^\s*(?:int32_t|int)\s+(gg_\w+)\s*\(', hdr, flags=re.M))
    assert declared == set(built.EXPORTS), (declared ^ set(built.EXPORTS))
    L = ctypes.CDLL(built.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert built.lib().gg_version() == built.ABI_VERSION == 5


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md is the binding contract a reference maintainer reads: it has to name every exported symbol."""
    hdr = open(os.path.join(ROOT, 'include', 'gymgo_amd.h')).read()
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    declared = set(re.findall(r'^\s*(?:int32_t|int)\s+(gg_\w+)\s*\(', hdr, flags=re.M))
    assert declared and not [n for n in sorted(declared) if n not in doc]


def test_argument_validation_without_device(built):
    L = built.lib()
    # argument checks come before any device work
    assert L.gg_batch_next_states(None, None, None, None, 4, 1, 0, None) == -1
    assert L.gg_batch_next_states(None, None, None, None, 4, 20, 0, None) == -1
    assert L.gg_batch_next_states(None, None, None, None, -1, 9, 0, None) == -1
    assert L.gg_batch_next_states(None, None, None, None, 0, 9, 0, None) == 0
    assert L.gg_batch_next_states(None, None, None, None, 2, 9, 0, None) == -2
    assert L.gg_batch_next_states_ws(None, None, None, None, None, 2, 9, 0, None) == -2
    assert L.gg_batch_next_states_ws(None, None, None, None, None, 2, 20, 0, None) == -1
    assert L.gg_batch_rollout(None, None, None, None, 2, 9, -1, 1, None) == -3
    assert L.gg_batch_children(None, None, 0, 19, 0, None) == 0
    assert L.gg_batch_env_step(None, None, None, None, None, None, None, 2, 9, 0.0, 0, 1, None) == -2
    assert L.gg_batch_env_step(None, None, None, None, None, None, None, 2, 9, 0.0, 7, 1, None) == -3
    assert L.gg_batch_env_step_packed(None, None, None, None, None, None, None, 2, 9, 0.0, 1, 1, None) == -2
    assert L.gg_batch_next_states_packed(None, None, None, None, 2, 1, 0, None) == -1
    assert L.gg_batch_rollout_packed(None, None, None, None, 2, 9, -4, 1, None) == -3
    assert L.gg_batch_children_packed(None, None, 3, 9, 0, None) == -2
    assert L.gg_packed_words(19) == 58 and L.gg_packed_words(1) == -1
    assert L.gg_batch_play_moves(None, None, None, 2, 9, -1, None) == -3
    assert L.gg_batch_play_moves_packed(None, None, None, 2, 9, 5, None) == -2


def test_no_cpu_fallback(built):
    """Host tensors / NumPy input without a device must fail loudly, never compute on the CPU."""
    import torch
    from gymgo_amd import gogame
    if torch.cuda.is_available():
        pytest.skip('device present')
    with pytest.raises(built.GymGoNativeError):
        gogame.next_state(np.zeros((6, 7, 7)), 3)
    with pytest.raises(built.GymGoNativeError):
        gogame.batch_next_states(torch.zeros((2, 6, 7, 7), dtype=torch.uint8), [1, 2])


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'gymgo_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.replace('oracle/gg_oracle.c', '').replace('mirrored by oracle', ''), f


def test_shard_partition():
    from gymgo_amd.envs.vec_env import shard
    for total, world in ((1048576, 8), (65536, 1), (10, 4), (7, 8)):
        spans = [shard(total, r, world) for r in range(world)]
        assert sum(c for _, c in spans) == total
        pos = 0
        for first, cnt in spans:
            assert first == pos
            pos += cnt
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_api_surface_matches_reference_names():
    """Same public names as gym_go/gogame.py + GoEnv methods."""
    from gymgo_amd import gogame, govars, state_utils
    from gymgo_amd.envs import GoEnv, make
    for name in ('init_state', 'batch_init_state', 'next_state', 'batch_next_states', 'invalid_moves', 'valid_moves',
                 'batch_invalid_moves', 'batch_valid_moves', 'children', 'action_size', 'prev_player_passed',
                 'batch_prev_player_passed', 'game_ended', 'batch_game_ended', 'winning', 'batch_winning', 'turn',
                 'batch_turn', 'liberties', 'num_liberties', 'areas', 'batch_areas', 'canonical_form',
                 'batch_canonical_form', 'random_symmetry', 'all_symmetries', 'random_weighted_action',
                 'random_action', 'str', 'batch_children'):
        assert callable(getattr(gogame, name)), name
    for name in ('compute_invalid_moves', 'batch_compute_invalid_moves', 'adj_data', 'batch_adj_data', 'set_turn',
                 'batch_set_turn'):
        assert callable(getattr(state_utils, name)), name
    assert (govars.BLACK, govars.WHITE, govars.TURN_CHNL, govars.INVD_CHNL, govars.PASS_CHNL, govars.DONE_CHNL,
            govars.NUM_CHNLS) == (0, 1, 2, 3, 4, 5, 6)
    for name in ('reset', 'step', 'state', 'canonical_state', 'children', 'valid_moves', 'uniform_random_action',
                 'turn', 'prev_player_passed', 'game_ended', 'winning', 'winner', 'reward', 'info', 'render', 'close'):
        assert callable(getattr(GoEnv, name)), name
    env = make('gym_go:go-v0', size=7, komi=2.5, reward_method='heuristic')
    assert env.size == 7 and env.komi == 2.5 and env.state().shape == (6, 7, 7)
    assert gogame.action_size(board_size=19) == 362
    with pytest.raises(RuntimeError):
        gogame.action_size()


def test_host_side_predicates_on_numpy():
    from gymgo_amd import gogame
    s = np.zeros((6, 5, 5))
    assert gogame.turn(s) == 0 and not gogame.prev_player_passed(s) and gogame.game_ended(s) == 0
    s[2] = 1; s[4] = 1; s[5] = 1
    assert gogame.turn(s) == 1 and gogame.prev_player_passed(s) and gogame.game_ended(s) == 1
    c = gogame.canonical_form(s)
    assert c[2].max() == 0 and s[2].max() == 1            # copy, input untouched
    b = np.stack([s, np.zeros((6, 5, 5))])
    b[0, 0, 1, 1] = 1
    cb = gogame.batch_canonical_form(b)
    assert cb[0, 1, 1, 1] == 1 and cb[0, 0, 1, 1] == 0 and cb[0, 2].max() == 0
    assert np.array_equal(cb, gogame.batch_canonical_form(cb))   # idempotent (test_batch_fns.py:14-34)
    assert list(gogame.batch_turn(b)) == [1, 0]
    assert len(gogame.all_symmetries(s)) == 8


def test_killed_group_listing_and_adj_table():
    """Host helpers of state_utils.update_pieces: killed mask -> per-group coordinate lists in raster order
    (the order scipy.ndimage.label + np.argwhere give the reference); the adj_locs go to the kernel as a padded
    table of flat indices - nothing is inferred about where a stone was placed."""
    from gymgo_amd import state_utils
    mask = np.zeros((5, 5), np.uint8)
    mask[0, 1] = mask[1, 0] = 1            # two single-stone groups
    mask[3, 2:5] = 1; mask[4, 4] = 1       # one L-shaped group
    groups = state_utils._killed_groups(mask)
    assert [g.tolist() for g in groups] == [[[0, 1]], [[1, 0]], [[3, 2], [3, 3], [3, 4], [4, 4]]]
    assert state_utils._killed_groups(np.zeros((3, 3))) == []
    s = np.zeros((6, 5, 5))
    s[1, 2, 2] = 1
    adj, surrounded = state_utils.adj_data(s, (2, 2), 1)
    assert sorted(map(tuple, adj)) == [(1, 2), (2, 1), (2, 3), (3, 2)] and not surrounded
    s[1, 0, 0] = 1
    adj2, _ = state_utils.adj_data(s, (0, 0), 1)
    assert sorted(map(tuple, adj2)) == [(0, 1), (1, 0)]
    table = state_utils._adj_table([adj, adj2, np.zeros((0, 2)), np.array([[4, 4], [9, 0], [0, -1], [1, 1], [2, 2], [3, 3]])], 5)
    assert table.dtype == np.int32 and table.shape == (4, 6)
    assert sorted(table[0][:4]) == [7, 11, 13, 17] and list(table[0][4:]) == [-1, -1]
    assert sorted(table[1][:2]) == [1, 5] and (table[1][2:] == -1).all() and (table[2] == -1).all()
    assert list(table[3]) == [24, -1, -1, 6, 12, 18]          # off-board locations are dropped, not wrapped
    s[0, 0, 1] = s[0, 1, 0] = 1
    assert state_utils.adj_data(s, (0, 0), 1)[1] is True


def test_host_side_argument_checks_without_device(built):
    """Shape / dtype / argument validation of the Python wrappers happens before any device work."""
    import torch
    from gymgo_amd import gogame
    with pytest.raises(ValueError):
        gogame._packed_size(torch.zeros((4, 57), dtype=torch.int32))      # 3N+1 with N=19 is 58
    with pytest.raises(ValueError):
        gogame._packed_size(torch.zeros((4, 58), dtype=torch.int64))
    assert gogame._packed_size(torch.zeros((4, 58), dtype=torch.int32)) == 19
    assert gogame._tracked_size(torch.zeros((2, 96), dtype=torch.int32)) == 19
    with pytest.raises(ValueError):
        gogame._tracked_size(torch.zeros((2, 58), dtype=torch.int32))
    assert gogame.packed_words(9) == 28 and gogame.tracked_words(9) == 46
    with pytest.raises(ValueError):
        gogame.batch_env_step(torch.zeros((2, 6, 5, 5), dtype=torch.uint8))              # neither actions nor rng
    with pytest.raises(KeyError):
        gogame.REWARD_METHODS['nope']
    if not torch.cuda.is_available():
        with pytest.raises(built.GymGoNativeError):                                         # host tensors never compute
            gogame.batch_rollout_packed(torch.zeros((2, 28), dtype=torch.int32), torch.zeros(2, dtype=torch.int64), 3)
        with pytest.raises(built.GymGoNativeError):
            gogame.batch_track(torch.zeros((2, 6, 9, 9), dtype=torch.uint8))


def test_gym_registration_is_checked_not_assumed(tmp_path):
    """'go-v0' is shared with the reference package.  With a gym whose registry already holds the
    reference's entry point (and refuses to re-register), importing gymgo_amd.envs must WARN and record that the bare id
    is not ours, while the namespaced id and envs.make() still resolve to this package's GoEnv.  Run in subprocesses
    against the import stub of gym under oracle/ref_harness/stubs (gym itself is not in the image)."""
    import subprocess
    import sys
    stubs = os.path.join(ROOT, 'oracle', 'ref_harness', 'stubs')
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([stubs, ROOT]))
    clean = ("import gymgo_amd.envs as e; from gym.envs import registration as r; "
             "assert e.REGISTERED_WITH_GYM and e.GYM_REGISTRATION == {'go-v0': True, 'gymgo_amd/go-v0': True}, e.GYM_REGISTRATION; "
             "assert r.registry['go-v0'] == 'gymgo_amd.envs:GoEnv'")
    taken = ("import warnings; from gym.envs import registration as r; r.registry['go-v0'] = 'gym_go.envs:GoEnv'\n"
             "def strict(id, entry_point, **kw):\n"
             "    if id in r.registry: raise RuntimeError('Cannot re-register id: ' + id)\n"
             "    r.registry[id] = entry_point\n"
             "r.register = strict\n"
             "with warnings.catch_warnings(record=True) as w:\n"
             "    warnings.simplefilter('always'); import gymgo_amd.envs as e\n"
             "assert e.GYM_REGISTRATION == {'go-v0': False, 'gymgo_amd/go-v0': True}, e.GYM_REGISTRATION\n"
             "assert any('go-v0' in str(x.message) and issubclass(x.category, RuntimeWarning) for x in w)\n"
             "assert r.registry['go-v0'] == 'gym_go.envs:GoEnv' and type(e.make('go-v0', size=5)).__module__ == 'gymgo_amd.envs.go_env'")
    for code in (clean, taken):
        res = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr[-1500:]


def test_cu_count_override_is_read_from_the_environment(built):
    """GYMGO_AMD_CUS=<n> (read once per process) sizes the grids for n compute units: gg_device_cus reports it without
    touching a device; values out of range are ignored (the device's own count - 0 without a GPU - is reported)."""
    import subprocess
    import sys
    code = ('import ctypes, sys; L = ctypes.CDLL(%r); L.gg_device_cus.restype = ctypes.c_int32; print(L.gg_device_cus())'
            % os.path.join(ROOT, 'gymgo_amd', 'libgymgo_amd.so'))

    def run(value):
        env = dict(os.environ)
        env.pop('GYMGO_AMD_CUS', None)
        if value is not None:
            env['GYMGO_AMD_CUS'] = value
        p = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr[-500:]
        return int(p.stdout.strip().splitlines()[-1])

    assert run('77') == 77 and run('4096') == 4096
    base = run(None)
    assert run('0') == base and run('-3') == base and run('5000') == base and run('many') == base
    assert run('12abc') == base and run('8 ') == base and run('') == base      # a typo is ignored, not read as 12 / 8
