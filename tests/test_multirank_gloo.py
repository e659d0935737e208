"""CPU, world_size 2 over gloo: the sharded rollout driver (the N>1 path of bench.py) - each rank owns a
contiguous slice of the game range, seeds it by GLOBAL game index, no data-path collective; the union of
the shards equals the single-rank run.  The device kernels are replaced by the oracle here (this is a
test of the sharding / seeding / reduction logic, which is all that differs between N=1 and N>1)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, size, plies, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from gymgo_amd.envs.vec_env import shard
    from oracle import c_oracle
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    first, count = shard(total, rank, world)
    rng = np.array([c_oracle.lib().gg_oracle_rng_seed(20260927, first + i) for i in range(count)], dtype=np.uint64)
    states = np.zeros((count, 6, size, size), np.uint8)
    dist.barrier()
    states, rng, _ = c_oracle.batch_rollout(states, rng, plies, True)
    dist.barrier()
    # the only cross-rank traffic of the N>1 path: scalar timing / step counters
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    steps = torch.tensor([count * plies], dtype=torch.int64)
    dist.all_reduce(steps, op=dist.ReduceOp.SUM)
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), states=states, first=first, tmax=t.numpy(), steps=steps.numpy())
    dist.destroy_process_group()


def test_two_rank_sharded_rollout(tmp_path):
    import torch.multiprocessing as mp
    from oracle import c_oracle
    total, size, plies, world = 37, 7, 60, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, size, plies, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / ('rank%d.npz' % r)) for r in range(world)]
    merged = np.concatenate([p['states'] for p in parts])
    assert [int(p['first']) for p in parts] == [0, 19]
    rng = c_oracle.rng_seed(20260927, total)
    want, _, _ = c_oracle.batch_rollout(np.zeros((total, 6, size, size), np.uint8), rng, plies, True)
    assert np.array_equal(merged, want)
    assert all(float(p['tmax'][0]) == 2.0 for p in parts)          # MAX over ranks reached every rank
    assert all(int(p['steps'][0]) == total * plies for p in parts)  # whole-job step count
