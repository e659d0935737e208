"""CPU, world_size 2 over gloo: bench.py's OWN per-rank driver (bench.run_rank: sharding by global game index,
de-synchronising burn-in, warm-up, barrier-fenced timed region, max-over-ranks time, played-steps reduction) with the
CPU oracle as the step backend - the union of the two ranks' shards must equal the single-rank run of the whole batch,
and the aggregate must count every step of every rank.  The N > 1 self-spawn path of `python bench.py --gpus N` is
exercised too (argument plumbing only: it needs GPUs to go further)."""
import hashlib
import json
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleBackend:
    """bench.py's backend interface on the C oracle (host arrays): what HipBackend does on the device."""
    name = 'oracle'

    def setup(self, count, size, first_game):
        from oracle import c_oracle
        import bench
        self.c, self.count, self.size, self.first = c_oracle, count, size, first_game
        self.states = np.zeros((count, 6, size, size), np.uint8)
        self.rng = np.array([c_oracle.lib().gg_oracle_rng_seed(bench.SEED, first_game + b) for b in range(count)], np.uint64)
        self.steps = 0

    def rollout(self, plies, lo=0, hi=None, count_steps=True):
        hi = self.count if hi is None else hi
        st, rg, _ = self.c.batch_rollout(self.states[lo:hi], self.rng[lo:hi], plies, True)
        self.states[lo:hi], self.rng[lo:hi] = st, rg
        if count_steps and lo == 0 and hi == self.count:
            self.steps += plies * self.count

    def sync(self):
        pass

    def played(self):
        return self.steps

    def timer(self):
        import time
        t = [0.0, 0.0]
        return (lambda: t.__setitem__(0, time.perf_counter())), (lambda: t.__setitem__(1, time.perf_counter())), \
            (lambda: (t[1] - t[0]) * 1e3)

    def comm_tensor(self, values):
        import torch
        return torch.tensor(values, dtype=torch.float64)


OPTS = {'size': 9, 'plies_per_step': 5, 'steps': 3, 'warmup': 1, 'games_per_gpu': 24, 'desync': 32, 'burn_in_steps': 1}


def _worker(rank, world, port, outdir):
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank), 'WORLD_SIZE': str(world)})
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import bench
    dist.init_process_group('gloo', rank=rank, world_size=world)
    back = OracleBackend()
    res = bench.run_rank(rank, world, back, dict(OPTS, world=world), dist)
    np.save(os.path.join(outdir, 'states_%d.npy' % rank), back.states)
    np.save(os.path.join(outdir, 'rng_%d.npy' % rank), back.rng)
    if rank == 0:
        json.dump(res, open(os.path.join(outdir, 'res.json'), 'w'))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_bench_run_rank_two_ranks_over_gloo(tmp_path):
    import torch.multiprocessing as mp
    import bench
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = json.load(open(tmp_path / 'res.json'))
    total = OPTS['games_per_gpu'] * world
    assert res['total_games'] == total and res['count'] == OPTS['games_per_gpu'] and res['first'] == 0
    assert res['steps_played'] == OPTS['steps'] * OPTS['plies_per_step'] * total           # summed over BOTH ranks
    assert res['value'] == pytest.approx(res['steps_played'] / res['wall_s']) and res['wall_s'] > 0
    # the communicator's own evidence: its backend and world size, the ranks an all-reduce of ones counted, one launch
    # time per rank (each written by its own rank into its slot)
    assert res['comm'] == {'backend': 'gloo', 'world_size': 2, 'ranks_counted': 2}
    assert len(res['per_rank_launch_ms']) == 2 and all(x > 0 for x in res['per_rank_launch_ms'])
    # every rank about itself, gathered through the communicator: DISTINCT shard offsets, and the steps of ITS shard
    per = res['per_rank']
    assert [p['first_game'] for p in per] == [bench.shard(total, r, world)[0] for r in range(world)] == [0, 24]
    assert [p['steps_played'] for p in per] == [OPTS['steps'] * OPTS['plies_per_step'] * OPTS['games_per_gpu']] * 2
    assert sum(p['steps_played'] for p in per) == res['steps_played']
    # the union of the shards == bench's own driver at world 1 over the whole batch with the same schedule per game:
    # rank r's games get the schedule of a 24-game shard (de-sync slices are per shard), so rebuild it shard by shard
    got = np.concatenate([np.load(tmp_path / ('states_%d.npy' % r)) for r in range(world)])
    got_rng = np.concatenate([np.load(tmp_path / ('rng_%d.npy' % r)) for r in range(world)])
    parts, parts_rng = [], []
    for r in range(world):
        first, count = bench.shard(total, r, world)
        solo = OracleBackend()
        solo.setup(count, OPTS['size'], first)           # the same global game indices, in one process
        chunk = (count + 15) // 16
        for g in range(1, 16):
            lo, hi = g * chunk, min(count, (g + 1) * chunk)
            if lo < hi:
                solo.rollout(g * OPTS['desync'] // 16, lo, hi)
        solo.rollout(OPTS['plies_per_step'] * (OPTS['burn_in_steps'] + OPTS['warmup'] + OPTS['steps']))
        parts.append(solo.states)
        parts_rng.append(solo.rng)
    assert np.array_equal(got, np.concatenate(parts)) and np.array_equal(got_rng, np.concatenate(parts_rng))
    assert hashlib.sha256(got.tobytes()).hexdigest() != hashlib.sha256(np.zeros_like(got).tobytes()).hexdigest()


def test_bench_shard_is_the_package_shard():
    import bench
    from gymgo_amd.envs.vec_env import shard
    for total, world in ((1048576, 8), (65536, 1), (100, 3), (7, 8)):
        parts = [bench.shard(total, r, world) for r in range(world)]
        assert parts == [shard(total, r, world) for r in range(world)]
        assert parts[0][0] == 0 and sum(c for _, c in parts) == total
        assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(world - 1))


def test_bench_argument_plumbing():
    import bench
    a = bench.parse_args(['--gpus', '8', '--steps', '20', '--warmup', '5'])
    assert (a.gpus, a.steps, a.warmup, a.plies_per_step) == (8, 20, 5, 256)       # F is NOT tied to --steps
    a = bench.parse_args(['--fuse', '64'])
    assert a.plies_per_step == 64 and a.gpus == 1
    assert bench.algo_bytes_per_step(19) == 4336 and bench.fused_bytes_per_game(19) == 4348
    # (round 6: full-size 19x19 launches of >= 8 plies on more than 128 games per CU take the thirty-two-board kernel, gg_v5.h)
    assert bench.rollout_kernel_name(19, 65536, 256, 256) == 'k_rollout5<19, 0>' and bench.rollout_kernel_name(19, 131072, 8, 256) == 'k_rollout5<19, 0>'
    assert bench.rollout_kernel_name(19, 32769, 256, 256) == 'k_rollout5<19, 0>'
    assert bench.rollout_kernel_name(19, 32768, 256, 256) == 'k_rollout4<19, 0, false, true, false, false>'
    assert bench.rollout_kernel_name(19, 65536, 7, 256) == 'k_rollout4<19, 0, false, true, false, false>'
    assert bench.rollout_kernel_name(18, 65536, 256, 256) == 'k_rollout4<19, 0, false, false, false, false>'
    assert bench.rollout_symbol_prefix('k_rollout5<19, 0>') == '_ZN2gg10k_rollout5ILi19ELi0EEE'
    assert bench.rollout_kernel_name(9, 4096, 256, 256) == 'k_rollout_lat<9, true, true, 0>'      # config 2 (use_lat: <= 64 games per CU, >= 3 plies)
    assert bench.rollout_kernel_name(9, 4096, 2, 256) == 'k_rollout2_w4<9, true>' and bench.rollout_kernel_name(9, 8194, 1, 256) == 'k_rollout2<9, true, false, true>'
    assert bench.rollout_kernel_name(19, 4096, 1, 256) == 'k_rollout2_w4<19, true>' and bench.rollout_kernel_name(19, 4098, 1, 256) == 'k_rollout2<19, true, false, true>'
    assert bench.rollout_kernel_name(9, 32769, 256, 256) == 'k_rollout4<9, 0, false, true, false, false>' and bench.rollout_kernel_name(9, 32768, 256, 256).startswith('k_rollout_lat<9')
    assert bench.rollout_kernel_name(13, 8192, 4, 256) == 'k_rollout_lat<13, true, true, 0>'
    assert bench.rollout_kernel_name(19, 7936, 64, 256) == 'k_rollout_lat<19, true, true, 0>' and bench.rollout_kernel_name(19, 7937, 64, 256).startswith('k_rollout2<19')
    assert bench.rollout_kernel_name(19, 2048, 8, 256).startswith('k_rollout_lat<19') and bench.rollout_kernel_name(19, 2048, 7, 256).startswith('k_rollout2<19')
    assert bench.rollout_kernel_name(13, 4096, 3, 256).startswith('k_rollout_lat<13') and bench.rollout_kernel_name(13, 4095, 3, 256).startswith('k_rollout2<13')
    assert bench.rollout_symbol_prefix('k_rollout_lat<9, true, true, 0>') == '_ZN2gg13k_rollout_latILi9ELb1ELb1ELi0ELb0EEE'
    assert bench.rollout_kernel_name(19, 65536, 1, 256) == 'k_env_step16<19, false>'        # (gg_kernels.hip: use_ns16, 3 groups per SIMD)
    assert bench.rollout_kernel_name(19, 32768, 1, 256) == 'k_rollout2<19, true, false, true>'
    assert bench.rollout_kernel_name(9, 16384, 1, 256) == 'k_env_step16<9, false>'
    assert bench.rollout_kernel_name(7, 65536, 1, 256) == 'k_rollout2<9, true, false, false>'


def test_speed_calibration_file():
    """The CPU baseline's port-vs-reference ratio comes from a committed calibration (oracle/ref_harness/pin_oracle.py --speed
    against the real reference: >= 3 repetitions x >= 2 000 positions), not from constants in bench.py."""
    import json
    import bench
    rec = json.load(open(os.path.join(ROOT, 'oracle', 'ref_harness', 'speed_calibration.json')))
    assert rec['repetitions'] >= 3 and rec['positions'] >= 2000
    for key in ('reference_steps_per_s', 'port_steps_per_s', 'port_vs_reference_speed'):
        assert 0 < rec[key]['min'] <= rec[key]['mean'] <= rec[key]['max'], key
    assert rec['host'] and rec['date'] and 'pin_oracle.py' in rec['generated_by']
    assert bench.speed_calibration() == rec
    assert bench._two_digits(40871.3) == 41000.0 and bench._two_digits(0) == 0
    assert not hasattr(bench, 'PORT_STEPS_PER_S_PER_CORE_BUILD_BOX')


def test_kernel_hash_masks_every_pc_relative_literal():
    """bench._mask_pc_relative: the `symbol - pc` literals after EVERY s_getpc_b64 are zeroed - also when two sites are only
    seven words apart (the scan of one site must stop at the next) - and nothing else is."""
    import struct
    import bench

    def site(lo):   # s_getpc_b64 s[0:1]; s_add_u32 s0, s0, <lit>; s_addc_u32 s1, s1, <lit>; a 2-word load
        return [0xBE801C00, 0x8000FF00, lo, 0x8201FF01, 0xFFFFFFFF, 0xDC508000, 0x027F0000]

    def code(offsets, tail):
        w = [0x7E000280]
        for o in offsets:
            w += site(o)
        return struct.pack('<%dI' % (len(w) + 1), *(w + [tail]))

    a = bench._mask_pc_relative(code([0xFFFFDAD8, 0xFFFFDB00, 0xFFFFDB34, 0xFFFFDB68], 0xBF810000))
    b = bench._mask_pc_relative(code([0xFFFFD418, 0xFFFFD440, 0xFFFFD474, 0xFFFFD4A8], 0xBF810000))
    assert a == b and len(a) == 4 * (1 + 4 * 7 + 1)
    assert bench._mask_pc_relative(code([1, 2, 3, 4], 0xBF810000)) != bench._mask_pc_relative(code([1, 2, 3, 4], 0xBF800000))   # an instruction differs
