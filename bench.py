#!/usr/bin/env python
"""bench.py - env steps/sec across batched games, 19x19 uniform-random rollouts (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = ONE PLY FOR EVERY GAME of the batch (B env transitions): sample a uniform valid action per
game on the device, apply it (capture resolution, ko, new invalid-move mask, turn flip), auto-reset
finished games.  Steps are issued as launches of gg_batch_rollout with `--fuse F` plies per launch
(F = 1: the state makes a full HBM round trip every ply, the per-ply vector-env path; F > 1: the
board stays on-chip for F plies).  K is rounded up to a multiple of F.  Inputs are resident in HBM
before the timed region; nothing but the kernel launches sits inside it.

Multi-GPU: the game batch is sharded across ranks, no data-path collective (games never interact);
only the barrier and the max-over-ranks timing use RCCL.  scaling = "weak" (fixed games per GPU).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_STEP = {19: 4336, 13: 2032, 9: 976, 7: 592}  # read 6N^2 + write 6N^2 + 4 B action
HBM_PEAK_GBS = 8000.0                                        # MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def _cpu_worker(args):
    size, seconds, seed = args
    sys.path.insert(0, ROOT)
    from oracle import np_oracle
    return np_oracle.random_rollout_steps(size, seconds, seed)


def cpu_baseline(size, cpu_seconds_total=20.0):
    """The NumPy/SciPy port of the reference's next_state (oracle/np_oracle.py, same SciPy calls per step,
    pinned bit-exact and speed-calibrated against the real reference) on this box's host cores:
    one worker per core, each plays uniform-random games with auto-reset for a fixed wall time."""
    import multiprocessing as mp
    cores = min(os.cpu_count() or 1, 16)
    seconds = max(2.0, cpu_seconds_total / cores)
    ctx = mp.get_context('spawn')
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        steps = pool.map(_cpu_worker, [(size, seconds, 1000 + i) for i in range(cores)])
    wall = time.perf_counter() - t0
    # the build's own C restatement (oracle/gg_oracle.c: bitboard-free flood fills, one thread) - reported, not the baseline
    c_rate = None
    try:
        import numpy as np
        from oracle import c_oracle
        st = np.zeros((32, 6, size, size), np.uint8)
        rg = c_oracle.rng_seed(7, 32)
        st, rg, _ = c_oracle.batch_rollout(st, rg, 300, True)     # into the middle game
        c0 = time.perf_counter()
        st, rg, _ = c_oracle.batch_rollout(st, rg, 600, True)
        c_rate = round(32 * 600 / (time.perf_counter() - c0), 1)
    except Exception:
        c_rate = None
    return {
        'value': round(sum(steps) / seconds, 1), 'unit': 'env steps/s', 'cores': cores, 'kind': 'port',
        'per_core': round(sum(steps) / seconds / cores, 1),
        'c_restatement_steps_per_s_one_core': c_rate,
        'sample': '%d workers x %.1f s of %dx%d uniform-random self-play with auto-reset (oracle/np_oracle.py, '
                  'same scipy.ndimage calls per step as the reference); %d steps total, pool wall %.1f s'
                  % (cores, seconds, size, size, sum(steps), wall),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1024)
    ap.add_argument('--warmup', type=int, default=128)
    ap.add_argument('--size', type=int, default=19)
    ap.add_argument('--games-per-gpu', type=int, default=0, help='0 = 65536 at 1 GPU, 131072 per GPU otherwise')
    ap.add_argument('--fuse', type=int, default=int(os.environ.get('GG_BENCH_FUSE', '256')),
                    help='plies per kernel launch')
    ap.add_argument('--burn-in', type=int, default=256, help='untimed plies before warmup (stationary board mix)')
    ap.add_argument('--desync', type=int, default=640, help='spread of extra burn-in plies across the batch (0 = lock-step)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=32.0)
    ap.add_argument('--no-also', action='store_true', help='skip the untimed per-ply extras (clean profiling passes)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d for --gpus %d' % (args.gpus, args.gpus))
        args.gpus = world

    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.size, args.cpu_seconds)   # before the GPU context exists (spawned workers)

    import torch
    import torch.distributed as dist
    from gymgo_amd import _lib, gogame
    from gymgo_amd.envs.vec_env import shard

    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)

    per_gpu = args.games_per_gpu or (65536 if world == 1 else 131072)
    total_games = per_gpu * world
    first, count = shard(total_games, rank, world)
    N, F = args.size, max(1, args.fuse)
    K, W = max(1, args.steps), max(0, args.warmup)    # EXACTLY K timed and W warm-up steps: full launches of F plies + a remainder
    F = min(F, K)

    def run_plies(n):
        for _ in range(n // F):
            gogame.batch_rollout(states, rng, F, True, None, steps_done)
        if n % F:
            gogame.batch_rollout(states, rng, n % F, True, None, steps_done)

    states = gogame.batch_init_state(count, N, device=dev)
    rng = gogame.rng_seed(count, 20260927, first, dev)
    steps_done = torch.zeros(count, dtype=torch.int64, device=dev)
    # De-synchronise the games first: slice g of 16 plays g * desync/16 extra plies, so that the batch holds every
    # game phase at once (stationary mix: mean game length of uniform-random 19x19 play is ~640 plies) and the
    # timed window does not depend on where a lock-step batch happens to be.  Untimed, and not counted in steps_done.
    if args.desync:
        chunk = (count + 15) // 16
        for gslice in range(1, 16):
            lo, hi = gslice * chunk, min(count, (gslice + 1) * chunk)
            if lo < hi:
                gogame.batch_rollout(states[lo:hi], rng[lo:hi], gslice * args.desync // 16, True)
    for _ in range((args.burn_in + F - 1) // F):   # same launch shape as the timed ones (rocprof averages then agree)
        gogame.batch_rollout(states, rng, F, True, None, steps_done)
    run_plies(W)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    before = int(steps_done.sum())
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fence()
    t0 = time.perf_counter()
    ev0.record()            # launches go to torch's current stream (gymgo_amd/_lib.py: stream_ptr)
    run_plies(K)
    ev1.record()
    fence()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1)
    played = int(steps_done.sum()) - before
    assert played == K * count, 'work was skipped inside the timed region (%d != %d)' % (played, K * count)

    # Untimed extras (outside the K timed steps): the same games through the per-ply paths, rank 0 only.
    also = {}
    if rank == 0 and not args.no_also:
        def rate(fn, reps):
            fn()
            torch.cuda.synchronize(dev)
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            for _ in range(reps):
                fn()
            a1.record()
            torch.cuda.synchronize(dev)
            return count * reps / (a0.elapsed_time(a1) * 1e-3)
        also['rollout_1_ply_per_launch_steps_per_s'] = round(rate(lambda: gogame.batch_rollout(states, rng, 1, True), 32), 1)
        acts = gogame.batch_sample_actions(states, rng)
        also['gg_batch_next_states_steps_per_s'] = round(
            rate(lambda: gogame.batch_next_states(states, acts, check=False), 16), 1)
        also['gg_batch_env_step_steps_per_s'] = round(
            rate(lambda: gogame.batch_env_step(states, None, rng, 7.5, 'real', True), 16), 1)
        if args.size == 19 and count >= 8192:   # BASELINE config 5: 8 192 mid-game parents, padded 362-slot expansion
            parents = states[:8192]
            kids = torch.empty((8192, args.size ** 2 + 1, 6, args.size, args.size), dtype=torch.uint8, device=dev)
            lib, n = _lib.lib(), args.size

            def expand():
                _lib.check(lib.gg_batch_children(_lib.dev_ptr(parents, torch.uint8, 'states'), _lib.dev_ptr(kids, torch.uint8, 'children'),
                                                 8192, n, 0, _lib.stream_ptr(dev)), 'gg_batch_children')
            per_s = rate(expand, 8) * 8192 / count
            also['gg_batch_children_parents_per_s'] = round(per_s, 1)
            also['gg_batch_children_write_roofline_frac'] = round(per_s * 786258 / 8e12, 4)
            del kids
        also['note'] = ('same resident batch; kernel-event time of 1-ply launches / of the out-of-place step API / of the fused '
                        'GoEnv.step (sample + step + areas + reward) / of the 362-slot children expansion of 8 192 parents')
        if world == 1 and count == 65536 and not args.games_per_gpu:
            # the per-GPU batch of the N > 1 lines (131 072 games, BASELINE config 4) on ONE GPU: the base for weak-scaling
            # ratios - 65 536 games fill the resident waves 1.33 times, 131 072 games 2.67 times
            big = 131072
            st2 = gogame.batch_init_state(big, N, device=dev)
            rg2 = gogame.rng_seed(big, 20260927, 0, dev)
            if args.desync:
                chunk2 = big // 16
                for gslice in range(1, 16):
                    gogame.batch_rollout(st2[gslice * chunk2:(gslice + 1) * chunk2], rg2[gslice * chunk2:(gslice + 1) * chunk2],
                                         gslice * args.desync // 16, True)
            gogame.batch_rollout(st2, rg2, F, True)
            torch.cuda.synchronize(dev)
            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b0.record()
            for _ in range(4):
                gogame.batch_rollout(st2, rg2, F, True)
            b1.record()
            torch.cuda.synchronize(dev)
            also['rollout_131072_games_steps_per_s'] = round(big * 4 * F / (b0.elapsed_time(b1) * 1e-3), 1)
            del st2, rg2

    t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t[0])
    if rank == 0:
        value = K * total_games / wall_max
        algo = ALGO_BYTES_PER_STEP.get(N, 12 * N * N + 4)
        launch_ms = kernel_ms * F / K     # per F plies (K is a multiple of F by default: then the average launch time)
        achieved = algo * count * F / (launch_ms * 1e-3) / 1e9
        traffic = None
        tj = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
        if os.path.exists(tj):
            try:
                rec = json.load(open(tj))
                if rec.get('size') == N and rec.get('fuse') == F and rec.get('games') == count:
                    traffic = rec.get('bytes_per_launch')
            except Exception:
                traffic = None
        rcap = 9 if N <= 9 else 13 if N <= 13 else 19
        if os.environ.get('GG_KERNEL_VARIANT') == '1':
            kernel_name = 'k_rollout<%d>' % rcap
        elif (F >= int(os.environ.get('GG_V3_MIN', '2')) and os.environ.get('GG_ROLLOUT_V2') != '1'
              and (count >= 32 * torch.cuda.get_device_properties(dev).multi_processor_count or os.environ.get('GG_V3_NB'))):
            # 12 boards per wave, liberty classes carried across plies; <row capacity, byte-plane I/O, drawn moves, N == capacity>
            kernel_name = 'k_rollout3<%d, 0, false, %s>' % (rcap, 'true' if N == rcap else 'false')
        else:
            kernel_name = 'k_rollout2<%d, %s, false, %s>' % (rcap, 'true' if F <= 2 else 'false', 'true' if N == rcap else 'false')
        line = {
            'metric': 'env steps/sec across batched games, 19x19 uniform-random rollouts',
            'value': round(value, 1), 'unit': 'env steps/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': round(wall_max * 1e3 / K, 6), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
            'config': {
                'workload': '%dx%d, %d parallel games%s, uniform-random rollouts with auto-reset, %d plies per launch'
                            % (N, N, total_games, '' if world == 1 else ' (%d per GPU)' % per_gpu, F),
                'board': N, 'games': total_games, 'games_per_gpu': per_gpu, 'plies_per_launch': F,
                'burn_in_plies': args.burn_in, 'desync_plies': args.desync, 'sharding': 'batch split across ranks, no collective',
            },
            'roofline': {
                'bound': 'hbm', 'kernel': kernel_name,
                'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic,
                'algorithmic_bytes_per_step': algo, 'steps_per_launch': count * F,
                'launch_ms': round(launch_ms, 5),
                'note': ('achieved = algorithmic bytes of the per-ply path (read + write one board + action per step) / '
                         'launch time; the fused kernel keeps the boards on-chip for all plies of a launch, so it can '
                         'exceed what any per-ply streaming implementation could reach (frac > 1); `traffic` is the '
                         'HBM traffic it really causes per launch'),
            },
        }
        if cpu is not None:
            line['cpu_baseline'] = cpu
        line['also'] = also
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
