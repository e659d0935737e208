#!/usr/bin/env python
"""bench.py - env steps/sec across batched games, 19x19 uniform-random rollouts (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: the ranks are spawned here)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A bench "step" = ONE LAUNCH of gg_batch_rollout over the whole resident batch: `--plies-per-step` (default 256)
plies for every game - sample a uniform valid action per game on the device, apply it (capture resolution, ko, new
invalid-move mask, turn flip), auto-reset finished games - with the boards resident on-chip between the plies of
the launch.  K timed steps are exactly K launches (K x plies x games env steps); `value` = env steps per second.
Inputs are resident in HBM before the timed region; nothing but the kernel launches sits inside it, and the
number of env steps actually played is asserted from the kernel's own per-game counters.

Multi-GPU: the game batch is sharded across ranks by global game index, no data-path collective (games never
interact); only the barrier, the max-over-ranks time and the played-steps check use the process group (RCCL).
scaling = "weak" (fixed games per GPU).  Rank 0 prints ONE JSON line.

`run_rank()` is the per-rank driver; tests/test_multirank_gloo.py runs it at world_size 2 over gloo with the CPU
oracle as the step backend, so the sharding / timing / reduction code tested there is the code that runs on 8 GPUs.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = 'env steps/sec across batched games, 19x19 uniform-random rollouts'
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
SEED = 20260927


def algo_bytes_per_step(n):
    """one out-of-place transition reads 6 N^2 + writes 6 N^2 + a 4-byte action (4 336 B at 19x19)."""
    return 12 * n * n + 4


def fused_bytes_per_game(n):
    """What a fused launch must move per game whatever its ply count: the board in, the board out, the generator state
    in and out (6 N^2 + 6 N^2 + 16 B)."""
    return 12 * n * n + 16


# ------------------------------------------------------------------------------------------------ CPU baseline
def speed_calibration():
    """oracle/ref_harness/speed_calibration.json: the NumPy/SciPy port timed against the REAL reference on the same positions
    (written by `oracle/ref_harness/pin_oracle.py --speed` where /root/reference exists: >= 3 repetitions x >= 2 000
    positions, with its range).  The reference cannot travel to the GPU box, its ratio to the port can.  None if absent."""
    try:
        rec = json.load(open(os.path.join(ROOT, 'oracle', 'ref_harness', 'speed_calibration.json')))
        r = rec['port_vs_reference_speed']
        if not (0.0 < float(r['min']) <= float(r['mean']) <= float(r['max'])):
            return None
        return rec
    except Exception:
        return None


def _two_digits(x):
    """x rounded to two significant digits (the estimate of the reference on this box is no better than that)."""
    if not x:
        return x
    import math
    q = 10 ** (int(math.floor(math.log10(abs(x)))) - 1)
    return float(round(x / q) * q)


def _cpu_worker(args):
    size, seconds, seed = args
    sys.path.insert(0, ROOT)
    from oracle import np_oracle
    return np_oracle.random_rollout_steps(size, seconds, seed)


def cpu_baseline(size, seconds_per_worker=6.0, max_workers=None):
    """The NumPy/SciPy port of the reference's next_state (oracle/np_oracle.py: the same scipy.ndimage calls per step,
    pinned bit-exact and speed-calibrated against the real reference) on this box's host cores: one single-threaded
    worker process per USABLE core (sched_getaffinity; `max_workers` caps it), each playing uniform-random games with
    auto-reset for a fixed wall time.  Runs on rank 0 at every world size, before the GPU context / process group exist."""
    import multiprocessing as mp
    host = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = host
    try:      # a container may own fewer CPUs' worth of time than it can see: more workers than that only contend
        quota = open('/sys/fs/cgroup/cpu.max').read().split()
        cgroup = None if quota[0] == 'max' else float(quota[0]) / float(quota[1])
    except Exception:
        cgroup = None
    granted = usable if not cgroup else max(1, min(usable, int(cgroup + 0.999)))
    cores = max(1, min(granted, max_workers or granted))
    ctx = mp.get_context('spawn')
    one_thread = {k: '1' for k in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS', 'NUMEXPR_NUM_THREADS')}
    saved = {k: os.environ.get(k) for k in one_thread}
    os.environ.update(one_thread)          # inherited by the spawned workers: one worker = one core
    t0 = time.perf_counter()
    try:
        with ctx.Pool(cores) as pool:
            steps = pool.map(_cpu_worker, [(size, seconds_per_worker, 1000 + i) for i in range(cores)])
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    wall = time.perf_counter() - t0
    c_rate = None   # the build's own C restatement, one thread: reported next to the baseline, it is not the baseline
    try:
        import numpy as np
        from oracle import c_oracle
        st = np.zeros((32, 6, size, size), np.uint8)
        rg = c_oracle.rng_seed(7, 32)
        st, rg, _ = c_oracle.batch_rollout(st, rg, 300, True)
        c0 = time.perf_counter()
        st, rg, _ = c_oracle.batch_rollout(st, rg, 600, True)
        c_rate = round(32 * 600 / (time.perf_counter() - c0), 1)
    except Exception:
        c_rate = None
    total = float(sum(steps))
    cal = speed_calibration()
    if cal:
        r = cal['port_vs_reference_speed']
        ratio_rec = {'mean': r['mean'], 'min': r['min'], 'max': r['max'], 'positions': cal.get('positions'),
                     'repetitions': cal.get('repetitions'), 'date': cal.get('date'), 'host': cal.get('host'),
                     'source': 'oracle/ref_harness/speed_calibration.json (pin_oracle.py --speed against the real reference)'}
        # the port is FASTER than the reference it restates: divide by the ratio to estimate the reference on this box
        ref_est = {'mean': _two_digits(total / seconds_per_worker / r['mean']),
                   'range': [_two_digits(total / seconds_per_worker / r['max']), _two_digits(total / seconds_per_worker / r['min'])]}
    else:
        ratio_rec, ref_est = 'unknown (oracle/ref_harness/speed_calibration.json missing or malformed)', None
    return {
        'value': round(total / seconds_per_worker, 1), 'unit': 'env steps/s', 'cores': cores, 'kind': 'port',
        'host_cpu_count': host, 'usable_cpus': usable, 'cgroup_cpu_quota_cores': None if cgroup is None else round(cgroup, 2), 'per_core': round(total / seconds_per_worker / cores, 1),
        'c_restatement_steps_per_s_one_core': c_rate,
        'port_vs_reference_speed': ratio_rec,
        'reference_estimate_steps_per_s': ref_est,
        'sample': '%d worker processes (os.cpu_count() = %d, usable %d, cgroup quota %s cores) x %.1f s of %dx%d uniform-random self-play with '
                  'auto-reset (oracle/np_oracle.py, same scipy.ndimage calls per step as the reference); %d steps total, '
                  'pool wall %.1f s' % (cores, host, usable, 'none' if cgroup is None else '%.1f' % cgroup, seconds_per_worker,
                                        size, size, int(total), wall),
    }


# ------------------------------------------------------------------------------------------------ step backends
class HipBackend:
    """The product path: uint8 [B,6,N,N] device tensor + gg_batch_rollout through the C-ABI on torch's current stream."""
    name = 'hip'

    def __init__(self, device):
        import torch
        from gymgo_amd import gogame
        self.torch, self.gogame, self.device = torch, gogame, device

    def setup(self, count, size, first_game):
        t, g = self.torch, self.gogame
        self.count, self.size = count, size
        self.states = g.batch_init_state(count, size, device=self.device)
        self.rng = g.rng_seed(count, SEED, first_game, self.device)
        self.steps_done = t.zeros(count, dtype=t.int64, device=self.device)

    def rollout(self, plies, lo=0, hi=None, count_steps=True):
        hi = self.count if hi is None else hi
        if lo == 0 and hi == self.count:
            self.gogame.batch_rollout(self.states, self.rng, plies, True, None, self.steps_done if count_steps else None)
        else:
            self.gogame.batch_rollout(self.states[lo:hi], self.rng[lo:hi], plies, True)

    def sync(self):
        self.torch.cuda.synchronize(self.device)

    def played(self):
        return int(self.steps_done.sum())

    def timer(self):
        t = self.torch
        ev = (t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True))
        return (lambda: ev[0].record()), (lambda: ev[1].record()), (lambda: ev[0].elapsed_time(ev[1]))

    def comm_tensor(self, values):
        return self.torch.tensor(values, dtype=self.torch.float64, device=self.device)

    def clock_sampler(self):
        return ClockSampler(self.torch, self.device)

    def state_digest(self):
        import hashlib
        return hashlib.sha256(self.states.cpu().numpy().tobytes()).hexdigest()

    def device_identity(self):
        """(device index, 48-bit digest of what identifies the physical device): lets the gathered line show that the
        ranks really sat on distinct GPUs."""
        import hashlib
        props = self.torch.cuda.get_device_properties(self.device)
        ident = '%s|%s|%s' % (getattr(props, 'uuid', ''), getattr(props, 'pci_bus_id', ''), props.name)
        return int(self.device.index or 0), int(hashlib.sha256(ident.encode()).hexdigest()[:12], 16)


class ClockSampler:
    """Shader clock / board power of the run, sampled from a host thread through amdsmi (torch.cuda.clock_rate /
    power_draw) while the warm-up and timed launches are in flight: a VALU-bound number moves with the clock the box
    happens to hold (DESIGN 5 records a lease that ran 45 % slower), so the line says which clock it was measured at."""

    def __init__(self, torch, device, period_s=0.004):
        import threading
        self.torch, self.device, self.period = torch, device, period_s
        self.mhz, self.watts, self.error = [], [], None
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        t = self.torch
        while not self._stop.is_set():
            try:
                self.mhz.append(float(t.cuda.clock_rate(self.device)))
                try:
                    self.watts.append(float(t.cuda.power_draw(self.device)))
                except Exception:
                    pass
            except Exception as e:       # no amdsmi on this box: the line says so
                self.error = '%s: %s' % (type(e).__name__, str(e)[:120])
                return
            self._stop.wait(self.period)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join(timeout=2.0)

    def record(self):
        if not self.mhz:
            return {'sclk_mhz': None, 'note': 'amdsmi clock query unavailable (%s)' % (self.error or 'no samples')}
        m = sorted(self.mhz)
        rec = {'sclk_mhz': {'min': m[0], 'median': m[len(m) // 2], 'max': m[-1], 'samples': len(m)},
               'note': 'torch.cuda.clock_rate (amdsmi current gfx clock) polled from a host thread'}
        if self.watts:
            w = sorted(self.watts)
            # torch documents milliwatts, amdsmi's socket power comes back in watts on this stack: raw values, unit unresolved
            rec['power_raw'] = {'min': round(w[0], 1), 'median': round(w[len(w) // 2], 1), 'max': round(w[-1], 1),
                                'unit': 'as returned by torch.cuda.power_draw (mW per its doc; W if amdsmi reports socket power)'}
        return rec


def pin_to_gpu_numa_node(torch, dev):
    """Pin this rank's host thread(s) to the CPUs of the NUMA node its GPU hangs off (the launches of one rank must not queue
    behind another socket's memory): the node from /sys/bus/pci/devices/<bdf>/numa_node, its CPUs from
    /sys/devices/system/node/node<k>/cpulist, intersected with what the process may use.  Returns a record for the line;
    never fails the run (a container may hide sysfs or refuse the affinity call)."""
    rec = {'numa_node': None, 'cpus_pinned': None}
    try:
        props = torch.cuda.get_device_properties(dev)
        bdf = getattr(props, 'pci_bus_id', None)
        if isinstance(bdf, int) or bdf is None:      # torch exposes domain / bus / device ids separately
            bdf = '%04x:%02x:%02x.0' % (getattr(props, 'pci_domain_id', 0), getattr(props, 'pci_bus_id', 0), getattr(props, 'pci_device_id', 0))
        node = int(open('/sys/bus/pci/devices/%s/numa_node' % str(bdf).lower()).read().strip())
        rec['numa_node'] = node
        if node < 0:
            rec['note'] = 'the platform reports no NUMA affinity for this GPU (numa_node = -1)'
            return rec
        cpus = set()
        for part in open('/sys/devices/system/node/node%d/cpulist' % node).read().strip().split(','):
            lo, _, hi = part.partition('-')
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = cpus & set(os.sched_getaffinity(0))
        if allowed:
            os.sched_setaffinity(0, allowed)
            rec['cpus_pinned'] = len(allowed)
        else:
            rec['note'] = 'no usable CPU on node %d: affinity left as it was' % node
    except Exception as e:
        rec['note'] = 'not pinned (%s: %s)' % (type(e).__name__, str(e)[:100])
    return rec


def shard(total_games, rank, world_size):
    """Contiguous equal split of the game index range [0, total_games) -> (first_game, count); the same function as
    gymgo_amd.envs.vec_env.shard (kept importable without torch for the CPU tests)."""
    base, rem = divmod(total_games, world_size)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def run_rank(rank, world, backend, opts, dist=None):
    """The per-rank bench driver.  backend: a step backend (HipBackend on the GPU; the gloo test passes one built on the
    CPU oracle).  dist: torch.distributed (already initialised) when launched as ranks, else None.  Returns the result record on rank 0
    (None elsewhere): value, wall time (max over ranks), kernel-event time of this rank, games and steps played."""
    N, F, K, W = opts['size'], opts['plies_per_step'], opts['steps'], opts['warmup']
    per_gpu = opts['games_per_gpu']
    total_games = per_gpu * world
    first, count = shard(total_games, rank, world)
    backend.setup(count, N, first)
    # De-synchronise the games first: slice g of 16 plays g * desync/16 extra plies, so that the batch holds every game
    # phase at once (stationary mix: the mean game length of uniform-random 19x19 play is ~640 plies) and the
    # timed window does not depend on where a lock-step batch happens to be.  Untimed, not counted.
    if opts['desync']:
        chunk = (count + 15) // 16
        for g in range(1, 16):
            lo, hi = g * chunk, min(count, (g + 1) * chunk)
            if lo < hi:
                backend.rollout(g * opts['desync'] // 16, lo, hi)
    for _ in range(opts['burn_in_steps']):      # same launch shape as the timed ones (rocprof averages then agree)
        backend.rollout(F, count_steps=False)
    # THE clock of the line, and a machine that has stopped ramping.  Untimed back-to-back launches of the timed shape run
    # BEFORE warm-up + timed region until the launch time has settled: windows of 8 launches (HIP events where the backend
    # has them), at least `clock_settle_s` (0.35 s) of them, until three consecutive windows agree within 0.5 % - capped at
    # `clock_settle_max_s` (3 s).  Why adaptive (round 6, tools/exp/timing_split.py, profiles/r06a_timing_split.txt): after an
    # idle gap the first ~10 launches of this shape run 5 % slow on one lease (1.96 -> 1.86 ms) and the driver's round-5 box
    # still ran 7 % slow after 0.35 s (1.98 ms timed, 1.85 ms for the same shape seconds later) - the clock ramp is per box.
    # The shader clock is sampled through amdsmi DURING this phase only (mean of the second half of the samples =
    # `clocks.sclk_mhz`); the timed region itself runs with no sampler thread (it cost 0.2 - 0.4 % there).
    settle, n_s = None, 0
    if hasattr(backend, 'clock_sampler') and opts.get('clock_settle_s', 0.35) > 0:      # (the CPU test backend has neither)
        probe = backend.clock_sampler()
        if probe is not None:
            probe.__enter__()
        t_s, windows = time.perf_counter(), []
        lo_s, hi_s = opts.get('clock_settle_s', 0.35), opts.get('clock_settle_max_s', 3.0)
        while True:
            w0, w1, w_ms = backend.timer()
            w0()
            for _ in range(8):
                backend.rollout(F, count_steps=False)
            w1()
            backend.sync()
            n_s += 8
            windows.append(w_ms() / 8)
            el = time.perf_counter() - t_s
            last = windows[-3:]
            flat = len(last) == 3 and max(last) - min(last) <= 0.005 * min(last)
            if (el >= lo_s and flat) or el >= hi_s:
                break
        if probe is not None:
            probe.__exit__(None, None, None)
        settle = {'sclk_mhz': None, 'launches': n_s, 'seconds': round(time.perf_counter() - t_s, 3),
                  'first_window_launch_ms': round(windows[0], 5), 'last_window_launch_ms': round(windows[-1], 5),
                  'settled': bool(flat),
                  'how': 'untimed back-to-back launches of the timed shape right before warm-up, in windows of 8, until three '
                         'windows agree within 0.5 %% (>= %.2f s, <= %.1f s); sclk = mean of the second half of the amdsmi '
                         'samples taken meanwhile; no sampler thread runs inside the timed region' % (lo_s, hi_s)}
        if probe is not None and probe.mhz:
            tail = probe.mhz[len(probe.mhz) // 2:]
            settle.update({'sclk_mhz': round(sum(tail) / len(tail), 1), 'samples': len(probe.mhz), 'min': min(probe.mhz),
                           'max': max(probe.mhz), 'power_raw': probe.record().get('power_raw')})
        elif probe is not None:
            settle['note'] = probe.record().get('note')
    # (the played-steps counter is read BEFORE the warm-up - whose launches do not count - so that nothing but the fence sits
    # between the last warm-up launch and the first timed one: a reduction + a host read there is an idle gap on the device)
    before = backend.played()
    for _ in range(W):
        backend.rollout(F, count_steps=False)

    def fence():
        backend.sync()
        if dist is not None:
            dist.barrier()
            backend.sync()

    start, stop, elapsed_ms = backend.timer()
    fence()
    t0 = time.perf_counter()
    start()
    for _ in range(K):
        backend.rollout(F)
    stop()
    fence()
    wall = time.perf_counter() - t0
    kernel_ms = elapsed_ms()
    played = backend.played() - before
    assert played == K * F * count, 'work was skipped inside the timed region (%d != %d)' % (played, K * F * count)
    red = backend.comm_tensor([wall, float(played)])
    comm = {'backend': None, 'world_size': 1, 'ranks_counted': 1}
    per_rank_ms = [kernel_ms / K]
    clocks = settle      # (None when the settle phase is switched off)
    sclk = (clocks or {}).get('sclk_mhz') or 0.0
    # what every rank reports about itself: the first global game index of its shard, the steps it played, the device it
    # ran on (index + a 48-bit digest of its uuid / name: two ranks on ONE device show up as equal pairs) and its clock
    dev_index, dev_tag = backend.device_identity() if hasattr(backend, 'device_identity') else (-1, 0)
    mine = [float(first), float(played), float(dev_index), float(dev_tag), float(sclk)]
    per_rank = [dict(zip(('first_game', 'steps_played', 'device_index', 'device_tag', 'sclk_mhz'), mine))]
    if dist is not None:
        tmax = red[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = red[1:].clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        wall_max, played_all = float(tmax[0]), int(tsum[0])
        # what the communicator itself saw: an all-reduce of ones counts the ranks, every rank's own launch time lands in
        # its slot of a zero vector (sum all-reduce = gather)
        slots = [0.0] * (world + 1)
        slots[rank], slots[world] = kernel_ms / K, 1.0
        gathered = backend.comm_tensor(slots)
        dist.all_reduce(gathered, op=dist.ReduceOp.SUM)
        per_rank_ms = [float(x) for x in gathered[:world]]
        comm = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(), 'ranks_counted': int(round(float(gathered[world])))}
        table = [0.0] * (world * len(mine))
        table[rank * len(mine):(rank + 1) * len(mine)] = mine
        table = backend.comm_tensor(table)
        dist.all_reduce(table, op=dist.ReduceOp.SUM)
        keys = ('first_game', 'steps_played', 'device_index', 'device_tag', 'sclk_mhz')
        per_rank = [dict(zip(keys, (float(x) for x in table[r * len(mine):(r + 1) * len(mine)]))) for r in range(world)]
    else:
        wall_max, played_all = wall, played
    # (dist None at world > 1: ONE rank of a bigger job driven on its own - tests/test_gpu_configs.py runs rank 5 of 8 that way)
    lone = dist is None and world > 1
    assert played_all == K * F * (count if lone else total_games)
    assert comm['ranks_counted'] == (1 if lone else world)
    for r, rec in enumerate(per_rank):      # every rank played its own shard, no two ranks the same one
        want_first, want_count = shard(total_games, rank if lone else r, world)
        assert int(rec['first_game']) == want_first and int(rec['steps_played']) == K * F * want_count, (r, rec)
        rec['first_game'], rec['steps_played'], rec['device_index'] = int(rec['first_game']), int(rec['steps_played']), int(rec['device_index'])
        rec['device_tag'] = '%012x' % int(rec['device_tag'])
    if rank != 0 and not lone:
        return None
    devs = [(rec['device_index'], rec['device_tag']) for rec in per_rank]
    return {'value': played_all / wall_max, 'wall_s': wall_max, 'kernel_ms': kernel_ms, 'total_games': total_games,
            'games_per_gpu': per_gpu, 'count': count, 'first': first, 'steps_played': played_all,
            'per_rank_launch_ms': per_rank_ms, 'per_rank': per_rank, 'distinct_devices': len(set(devs)) if devs[0][0] >= 0 else None,
            'comm': comm, 'clocks': clocks, 'settle_launches': n_s}


# ------------------------------------------------------------------------------------------------ roofline records
def device_cus():
    """The CU count the LIBRARY sizes its grids and picks its kernels for (the device's own, or GYMGO_AMD_CUS)."""
    from gymgo_amd import _lib
    return int(_lib.lib().gg_device_cus())


def rollout_kernel_name(n, games, plies, cus, auto_reset=True):
    """Mirror of gg_batch_rollout's dispatch (gymgo_amd/csrc/gg_kernels.hip: use_lat, use_multi_ply, use_ns16): which kernel
    serves this launch."""
    rcap = 9 if n <= 9 else 13 if n <= 13 else 19
    full = 'true' if n == rcap else 'false'
    lat_per_cu, lat_plies = {9: (128, 3), 13: (80, 3 if games >= 16 * cus else 4), 19: (31, 8)}[rcap]
    if plies >= lat_plies and games <= lat_per_cu * cus:
        return 'k_rollout_lat<%d, %s, %s, 0>' % (rcap, full, 'true' if auto_reset else 'false')
    if n == rcap and plies >= 8 and games > (128 if n == 19 else 159) * cus:
        return 'k_rollout5<%d, 0>' % n                          # a full machine: 32 boards per wave, flood jobs (gg_v5.h)
    if plies >= 2 and games >= 32 * cus:
        return 'k_rollout4<%d, 0, false, %s, false, false>' % (rcap, full)
    if plies == 1 and n in (9, 13, 19):
        # one ply per launch on a big batch of full-size boards: the sixteen-board env-step kernel from 3 / 2 / 1 groups of
        # sixteen boards per SIMD on (19x19 / 13x13 / 9x9)
        per_simd = {19: 3, 13: 2, 9: 1}[n]
        if (games + 15) // 16 >= cus * 4 * per_simd:
            return 'k_env_step16<%d, false>' % n
    if plies <= 2 and (games + 1) // 2 <= cus * (16 if n <= 9 else 8):
        return 'k_rollout2_w4<%d, %s>' % (rcap, full)        # small one- / two-ply launches: four waves per workgroup
    return 'k_rollout2<%d, %s, false, %s>' % (rcap, 'true' if plies <= 2 else 'false', full)


def kernel_code_hash(symbol_prefix, lib_path=None):
    """sha256[:16] of the machine code of the kernel whose mangled name starts with `symbol_prefix`, cut out of the gfx950
    code object inside the shared library (clang offload bundle -> AMDGPU ELF -> .symtab).  tools/summarize_profiles.py
    stores it next to the PMC record of a profile pass; bench.py recomputes it from the library it runs, so an edit of
    the kernel after the PMC pass shows up as `pmc_stale: true` instead of silently keeping the old instruction mix.
    None when the library or the symbol cannot be found."""
    import hashlib
    import struct
    path = lib_path or os.path.join(ROOT, 'gymgo_amd', 'libgymgo_amd.so')
    try:
        d = open(path, 'rb').read()
        magic = b'__CLANG_OFFLOAD_BUNDLE__'
        elfs, i = [], d.find(magic)
        while i >= 0:       # one bundle per translation unit (gg_kernels.hip, gg_rollout.hip)
            n = struct.unpack_from('<Q', d, i + len(magic))[0]
            p = i + len(magic) + 8
            for _ in range(n):
                off, size, tl = struct.unpack_from('<QQQ', d, p)
                p += 24
                triple = d[p:p + tl]
                p += tl
                if b'amdgcn' in triple and size and d[i + off:i + off + 4] == b'\x7fELF':
                    elfs.append(d[i + off:i + off + size])
            i = d.find(magic, i + 1)
        for elf in elfs:
            shoff, = struct.unpack_from('<Q', elf, 0x28)
            shentsize, shnum, shstrndx = struct.unpack_from('<HHH', elf, 0x3A)
            secs = [struct.unpack_from('<IIQQQQIIQQ', elf, shoff + k * shentsize) for k in range(shnum)]
            for sec in secs:
                if sec[1] != 2:      # SHT_SYMTAB
                    continue
                stroff = secs[sec[6]][4]
                for k in range(sec[5] // 24):
                    name_i, info, other, shndx, value, size = struct.unpack_from('<IBBHQQ', elf, sec[4] + 24 * k)
                    end = elf.index(b'\0', stroff + name_i)
                    name = elf[stroff + name_i:end]
                    if (info & 15) == 2 and size and name.startswith(symbol_prefix.encode()) and not name.endswith(b'.kd'):
                        tsec = secs[shndx]
                        code = elf[tsec[4] + (value - tsec[3]):tsec[4] + (value - tsec[3]) + size]
                        return hashlib.sha256(_mask_pc_relative(code)).hexdigest()[:16]
    except Exception:
        return None
    return None


def _mask_pc_relative(code):
    """The kernel's machine code with the PC-relative offsets of the translation unit's globals zeroed: after an s_getpc_b64
    the 32-bit literals of the s_add_u32 / s_addc_u32 that follow hold `symbol - pc` (the class-code table, the FairShare
    board), which move whenever ANY kernel of the unit changes size.  With them masked the hash says "this kernel's
    instructions", which is what a PMC record is about."""
    import struct
    n = len(code) // 4
    w = list(struct.unpack_from('<%dI' % n, code))
    i = 0
    while i < n:
        if (w[i] & 0xFF80FF00) == 0xBE801C00:                       # SOP1 s_getpc_b64
            j = i + 1
            while j < min(n - 1, i + 8):
                d = w[j]
                if (d & 0xFF80FF00) == 0xBE801C00:                   # the next s_getpc_b64 (sites can be 7 words apart): its own turn
                    break
                if (d & 0xFF800000) in (0x80000000, 0x82000000) and (((d >> 8) & 0xFF) == 0xFF or (d & 0xFF) == 0xFF):
                    w[j + 1] = 0                                     # s_add_u32 / s_addc_u32 with a 32-bit literal
                    j += 2
                else:
                    j += 1
            i = j
        else:
            i += 1
    return struct.pack('<%dI' % n, *w) + code[4 * n:]


def rollout_symbol_prefix(kernel):
    """Mangled-name prefix of a rollout kernel instantiation given as rollout_kernel_name writes it (k_rollout5<R, IO>, k_rollout4<R, IO, MOVES,
    FULLN, ENV, WTS>, k_rollout_lat<R, FULLN, AUTO, IO>, k_rollout2<R, PERPLY, PACKED, FULLN>)."""
    import re
    b = lambda x: 'Lb1E' if x == 'true' else 'Lb0E'
    m = re.match(r'k_rollout4<(\d+), (\d+), (\w+), (\w+), (\w+), (\w+)>', kernel)
    if m:
        return '_ZN2gg10k_rollout4ILi%sELi%sE%s%s%s%sEE' % (m.group(1), m.group(2), b(m.group(3)), b(m.group(4)), b(m.group(5)), b(m.group(6)))
    m = re.match(r'k_rollout5<(\d+), (\d+)>', kernel)
    if m:
        return '_ZN2gg10k_rollout5ILi%sELi%sEEE' % (m.group(1), m.group(2))
    m = re.match(r'k_rollout_lat<(\d+), (\w+), (\w+), (\d+)>', kernel)
    if m:
        return '_ZN2gg13k_rollout_latILi%sE%s%sLi%sELb0EEE' % (m.group(1), b(m.group(2)), b(m.group(3)), m.group(4))      # (+ SHORT = false)
    m = re.match(r'k_rollout2_w4<(\d+), (\w+)>', kernel)
    if m:
        return '_ZN2gg13k_rollout2_w4ILi%sE%sEE' % (m.group(1), b(m.group(2)))
    m = re.match(r'k_rollout2<(\d+), (\w+), (\w+), (\w+)>', kernel)
    if m:
        return '_ZN2gg10k_rollout2ILi%sE%s%s%sEE' % (m.group(1), b(m.group(2)), b(m.group(3)), b(m.group(4)))
    return None


def load_pmc(kernel, n, plies, games):
    """Instruction mix and HBM traffic of this exact launch shape from the committed PMC passes
    (profiles/pmc_rollout.json, written by tools/summarize_profiles.py from rocprofv3 --pmc runs of this command)."""
    path = os.path.join(ROOT, 'profiles', 'pmc_rollout.json')
    try:
        same_shape = None
        for rec in json.load(open(path)).get('records', []):
            if rec.get('kernel') == kernel and rec.get('size') == n and rec.get('plies_per_launch') == plies:
                if rec.get('games') == games:
                    return rec
                same_shape = rec
        if same_shape is not None:
            # another batch size of the same kernel and launch length (the multi-GPU lines run 131 072 games per GPU):
            # the instruction mix per env step does not depend on the batch, the traffic scales with the games
            rec = dict(same_shape)
            scale = games / float(same_shape['games'])
            for k in ('hbm_bytes_per_launch', 'fetch_bytes', 'write_bytes'):
                if rec.get(k) is not None:
                    rec[k] = int(round(rec[k] * scale))
            rec['source'] = '%s; measured at %d games, traffic scaled to %d' % (same_shape.get('source'), same_shape['games'], games)
            return rec
    except Exception:
        pass
    return None


def roofline_record(dev, n, games, plies, launch_ms, per_ply, clocks=None):
    import torch
    props = torch.cuda.get_device_properties(dev)
    cus = device_cus() or props.multi_processor_count      # what the library dispatches for (GYMGO_AMD_CUS included)
    clock_hz = float(getattr(props, 'clock_rate', 2400000)) * 1e3      # kHz -> Hz
    kernel = rollout_kernel_name(n, games, plies, cus)
    steps_per_launch = games * plies
    steps_per_s = steps_per_launch / (launch_ms * 1e-3)
    peak = cus * 4 * clock_hz / 2.0 / 1e9          # wave64 instructions per second, one per SIMD every 2 cycles
    pmc = load_pmc(kernel, n, plies, games)
    fused_bytes = fused_bytes_per_game(n) * games
    rec = {
        'bound': 'valu', 'kernel': kernel, 'unit': 'Gwave-instr/s', 'peak': round(peak, 2),
        'peak_note': '%d CUs x 4 SIMDs x %.2f GHz / 2 cycles per wave64 VALU instruction (the fastest ops: '
                     'profiles/r01_ubench_valu_rates.txt)' % (cus, clock_hz / 1e9),
        'launch_ms': round(launch_ms, 5), 'steps_per_launch': steps_per_launch, 'env_steps_per_s': round(steps_per_s, 1),
    }
    if pmc:
        ipe = pmc['instr_per_step']
        # the peak is a VALU-port peak, so only VALU instructions count against it; SALU and LDS instructions issue on
        # other ports and are reported beside it (round 3 added them into the numerator: 0.65 instead of 0.56)
        rec.update({
            'achieved': round(ipe['valu'] * steps_per_s / 1e9, 2), 'frac': round(ipe['valu'] * steps_per_s / 1e9 / peak, 4),
            'other_ports': {'salu_Ginstr_per_s': round(ipe['salu'] * steps_per_s / 1e9, 2), 'lds_Ginstr_per_s': round(ipe['lds'] * steps_per_s / 1e9, 2),
                            'note': 'not part of achieved / frac'},
            'instr_per_step': ipe, 'traffic': pmc.get('hbm_bytes_per_launch'), 'pmc_source': pmc.get('source'),
        })
        # is the PMC pass still about the code that runs?  (hash of the kernel's machine code in the loaded library)
        now = kernel_code_hash(rollout_symbol_prefix(kernel) or '\0')
        rec['kernel_code_sha16'] = now
        rec['pmc_kernel_code_sha16'] = pmc.get('kernel_code_sha16')
        rec['pmc_stale'] = (now is None or pmc.get('kernel_code_sha16') is None or now != pmc.get('kernel_code_sha16'))
        if rec['pmc_stale']:
            rec['pmc_stale_note'] = ('the instruction mix / traffic above were counted on ANOTHER build of this kernel (or the '
                                     'hash is missing): re-run tools/profile_round.sh')
        if kernel.startswith('k_rollout5<19'):
            rec['frac_note'] = ('frac prices INSTRUCTIONS against the issue port (VALU wave-instructions per env step x env steps/s / peak): '
                                'k_rollout5 issues %.1f per env step where the kernel of rounds 3 - 5 (k_rollout4) issued 73.2 - 74.1 at '
                                'frac 0.51 - 0.54 and 8.45 - 9.1e9 env steps/s; removing instructions lowers the fraction while the rate '
                                'rises (DESIGN.md 6).  Two waves per SIMD: a two-wave SIMD running flood visits alone issues one VALU '
                                'instruction per 1.47 ns (profiles/r06i_ubench_dep_chain2.txt)' % ipe['valu'])
            rec['env_steps_per_s_vs_round5_driver_value'] = round(steps_per_s / 8.45e9, 3)
        busy = pmc.get('valu_busy')
        if busy:
            rec['valu_busy_counters'] = busy      # counter-derived (tools/summarize_profiles.py), with its formula
        cpi = pmc.get('valu_issue_cycles_per_instr')
        if cpi:
            # a STATIC estimate (tools/isa_mix.py: the instruction mix of the ply loop priced with the measured issue
            # rates - about a third of the VALU instructions are 4-cycle ops), kept next to the counter-derived figure
            rec['valu_pipe_busy_static_estimate'] = round(ipe['valu'] * steps_per_s / 1e9 / peak * cpi / 2.0, 4)
            rec['valu_pipe_busy_static_note'] = ('frac x %.2f / 2 issue cycles per VALU instruction: %s'
                                                 % (cpi, pmc.get('valu_issue_cycles_source')))
    else:
        rec.update({'achieved': None, 'frac': None, 'traffic': None,
                    'pmc_source': 'no PMC pass for this launch shape under profiles/ (tools/profile_round.sh collects one)'})
    # the same fraction against the clock the run actually held (median of the amdsmi samples), next to the nominal one
    mhz = None
    if clocks and not isinstance(clocks.get('sclk_mhz'), dict):
        mhz = clocks.get('sclk_mhz')
    if mhz:
        peak_m = cus * 4 * mhz * 1e6 / 2.0 / 1e9
        rec['measured_clock'] = {'sclk_mhz': mhz, 'peak': round(peak_m, 2),
                                 'frac': round(rec['achieved'] / peak_m, 4) if rec.get('achieved') else None,
                                 'note': 'peak and frac recomputed with THE clock of the line (`clocks.sclk_mhz`: >= 0.3 s of '
                                         'back-to-back launches of the timed shape right before the timed region)'}
    else:
        rec['measured_clock'] = None
    rec['hbm'] = {
        'note': 'fused launch: board in + board out + generator per game per LAUNCH, whatever the ply count',
        'algorithmic_bytes_per_launch': fused_bytes, 'achieved': round(fused_bytes / (launch_ms * 1e-3) / 1e9, 2),
        'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(fused_bytes / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
        'traffic': pmc.get('hbm_bytes_per_launch') if pmc else None,
    }
    if per_ply:
        rec['per_ply'] = per_ply
    return rec


def children_record(torch, dev, parents, parents_note, by_phase=False, reps=8):
    """Config 5 - gg_batch_children on `parents` (uint8 [B,6,N,N]) into ONE preallocated buffer - and the un-padded form
    (gg_batch_children_offsets + gg_batch_children_compact) of the same parents into the same buffer: HIP events over
    back-to-back calls through the C-ABI.  THE harness for this number: bench.py's line and tools/bench_ops.py both call it
    (round 4 had two harnesses - another buffer, other parents, an allocation per call - that read 13 % apart on one lease)."""
    from gymgo_amd import _lib, gogame
    B, _, N, _ = parents.shape
    A, S = N * N + 1, 6 * N * N
    kids = torch.empty((B, A, 6, N, N), dtype=torch.uint8, device=dev)
    offs = torch.empty(B + 1, dtype=torch.int32, device=dev)
    order = torch.empty(B, dtype=torch.int32, device=dev)
    lib = _lib.lib()
    cur = {'p': parents}

    def expand():   # (reads cur['p'] at call time)
        _lib.check(lib.gg_batch_children(_lib.dev_ptr(cur['p'], torch.uint8, 'states'), _lib.dev_ptr(kids, torch.uint8, 'children'),
                                         B, N, 0, _lib.stream_ptr(dev)), 'gg_batch_children')

    def expand_compact():
        _lib.check(lib.gg_batch_children_offsets(_lib.dev_ptr(cur['p'], torch.uint8, 'states'), _lib.dev_ptr(offs, torch.int32, 'offsets'),
                                                 _lib.dev_ptr(order, torch.int32, 'order'), B, N, _lib.stream_ptr(dev)), 'gg_batch_children_offsets')
        _lib.check(lib.gg_batch_children_compact(_lib.dev_ptr(cur['p'], torch.uint8, 'states'), _lib.dev_ptr(offs, torch.int32, 'offsets'),
                                                 _lib.dev_ptr(order, torch.int32, 'order'), _lib.dev_ptr(kids, torch.uint8, 'children'),
                                                 B, N, 0, _lib.stream_ptr(dev)), 'gg_batch_children_compact')
    r, ms = event_rate(torch, dev, expand, B, reps)
    bytes_per_parent = S + A * S
    rec = {
        'parents': parents_note, 'harness': 'bench.children_record (one preallocated output buffer, HIP events over %d calls)' % reps,
        'parents_per_s': round(r, 1), 'child_states_per_s': round(r * A, 1), 'launch_ms': round(ms, 4),
        'roofline': {'bound': 'hbm', 'kernel': 'k_children3<%d, false, false>' % N, 'algorithmic_bytes_per_parent': bytes_per_parent,
                     'achieved': round(bytes_per_parent * r / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(bytes_per_parent * r / 1e9 / HBM_PEAK_GBS, 4)}}
    # HBM bytes of this launch from the committed counter passes (tools/calib_traffic.py under rocprofv3 --pmc), tied to
    # the kernel's machine code like the fused kernel's record
    try:
        crec = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_rollout.json'))).get('children') or {}
        chash = kernel_code_hash(CHILDREN_SYMBOL_PREFIX)
        stale = crec.get('kernel_code_sha16') != chash
        rec['roofline'].update({
            'traffic': None if stale else crec.get('hbm_bytes_per_launch'),
            'algorithmic_bytes_per_launch': bytes_per_parent * B, 'kernel_code_sha16': chash,
            'pmc_kernel_code_sha16': crec.get('kernel_code_sha16'), 'pmc_stale': stale, 'pmc_source': crec.get('source')})
    except Exception:
        rec['roofline']['traffic'] = None
    # the un-padded form (gogame.children(padded=False), gym_go/gogame.py:179) of the same parents: only the children
    # valid_moves() keeps, at their rank - both launches (offsets + children) inside the timed call
    rc, msc = event_rate(torch, dev, expand_compact, B, reps)
    total = int(offs[B].item())
    moved = total * S + B * (4 * N * N + 4) + 16 * B        # children written + planes 0-3 and flags read + offsets and order
    rec['compact'] = {
        'entry': 'gg_batch_children_offsets + gg_batch_children_compact (parents handed out by falling child count)', 'kernel': 'k_children3<%d, false, true>' % N,
        'parents_per_s': round(rc, 1), 'child_states_per_s': round(rc * total / B, 1), 'launch_ms': round(msc, 4),
        'mean_children_per_parent': round(total / B, 2), 'of_slots': A,
        'bytes_moved_per_launch': moved, 'bytes_vs_padded': round(moved / (bytes_per_parent * B), 4),
        'achieved': round(moved / (msc * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
        'frac': round(moved / (msc * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'speedup_vs_padded': round(ms / msc, 3)}
    if by_phase:
        # the same expansion for parents of ONE game phase each (the floods run over the empty points next to a stone,
        # the 362 slots are written whatever the position: the padded rate barely depends on the phase, the un-padded one does)
        phases = {}
        for phase, plies in (('early_20_plies', 20), ('mid_150_plies', 150), ('late_400_plies', 400)):
            ph = gogame.batch_init_state(B, N, device=dev)
            gogame.batch_rollout(ph, gogame.rng_seed(B, 77, 0, dev), plies, False)
            cur['p'] = ph
            rp, msp = event_rate(torch, dev, expand, B, 6)
            rcp, mscp = event_rate(torch, dev, expand_compact, B, 6)
            phases[phase] = {'parents_per_s': round(rp, 1), 'launch_ms': round(msp, 4),
                             'mean_stones': round(float((ph[:, 0] | ph[:, 1]).sum()) / B, 1),
                             'hbm_frac': round(bytes_per_parent * rp / 1e9 / HBM_PEAK_GBS, 4),
                             'compact_parents_per_s': round(rcp, 1), 'compact_launch_ms': round(mscp, 4),
                             'compact_mean_children': round(int(offs[B].item()) / B, 1)}
        rec['by_game_phase'] = phases
    del kids
    return rec


CHILDREN_SYMBOL_PREFIX = '_ZN2gg11k_children3ILi19ELb0ELb0EEE'      # k_children3<19, false, false>


def event_rate(torch, dev, fn, units, reps):
    """units/s by HIP events over `reps` back-to-back calls (after one untimed call)."""
    fn()
    torch.cuda.synchronize(dev)
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record()
    for _ in range(reps):
        fn()
    a1.record()
    torch.cuda.synchronize(dev)
    ms = a0.elapsed_time(a1) / reps
    return units / (ms * 1e-3), ms


def extras(dev, back, opts):
    """Untimed extras on rank 0 (outside the K timed steps): the per-ply paths on the same resident batch with their
    SURVEY 8(d) roofline fractions, and one driver-timed number for every other BASELINE config."""
    import torch
    from gymgo_amd import _lib, gogame
    from gymgo_amd.envs import make
    N, count = opts['size'], back.count
    states, rng = back.states, back.rng
    algo = algo_bytes_per_step(N)
    out, per_ply = {}, None
    # --- per-ply kernels, config-3 batch: out-of-place step API with caller-owned outputs (no allocation per call)
    acts = gogame.batch_sample_actions(states, rng)
    nxt, status = torch.empty_like(states), torch.empty(count, dtype=torch.int32, device=dev)
    r, ms = event_rate(torch, dev, lambda: gogame.batch_next_states(states, acts, check=False, out=nxt, status=status), count, 32)
    # (gg_kernels.hip: full-size boards take the sixteen-boards-per-wave kernel from four groups per SIMD on)
    big = N in (9, 13, 19) and (count + 15) // 16 >= (16 if N == 19 else 8) * int(_lib.lib().gg_device_cus())
    per_ply = {'kernel': ('k_next_states16<%d>' % N) if big else 'k_next_states2<%d>' % (9 if N <= 9 else 13 if N <= 13 else 19),
               'entry': 'gg_batch_next_states',
               'algorithmic_bytes_per_step': algo, 'launch_us': round(ms * 1e3, 2), 'env_steps_per_s': round(r, 1),
               'achieved': round(algo * r / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
               'frac': round(algo * r / 1e9 / HBM_PEAK_GBS, 4), 'bound': 'hbm',
               'note': 'through the Python API (ctypes call + launch), HIP events over 32 back-to-back calls'}
    out['gg_batch_next_states_steps_per_s'] = round(r, 1)
    # the same API as a rollout loop (every output is the next input, passes keep the loop stationary) with the
    # caller-owned workspace that carries the liberty classes from call to call (gg_batch_next_states_ws)
    wsp = gogame.next_states_workspace(count, N, dev)
    passes = torch.full((count,), N * N, dtype=torch.int32, device=dev)
    pp = [states.clone(), nxt]

    def loop_step():
        gogame.batch_next_states(pp[0], passes, check=False, out=pp[1], status=status, workspace=wsp)
        pp[0], pp[1] = pp[1], pp[0]
    loop_step()
    r, ms = event_rate(torch, dev, loop_step, count, 32)
    out['gg_batch_next_states_workspace_loop_steps_per_s'] = round(r, 1)
    out['gg_batch_next_states_workspace_loop_hbm_frac'] = round(algo * r / 1e9 / HBM_PEAK_GBS, 4)
    del nxt, wsp, pp
    r, _ = event_rate(torch, dev, lambda: gogame.batch_rollout(states, rng, 1, True), count, 32)
    out['rollout_1_ply_per_launch_steps_per_s'] = round(r, 1)
    env_out = (torch.empty(count, dtype=torch.float32, device=dev), torch.empty(count, dtype=torch.uint8, device=dev),
               torch.empty(count, dtype=torch.int32, device=dev), torch.empty(count, dtype=torch.int32, device=dev))
    r, _ = event_rate(torch, dev, lambda: gogame.batch_env_step(states, None, rng, 7.5, 'real', True, out=env_out), count, 32)
    out['gg_batch_env_step_byte_planes_steps_per_s'] = round(r, 1)
    # the batched env as GoVecEnv runs it: boards resident in the tracked format, one launch per step that also writes the
    # uint8 [B,6,N,N] observation (reads 384 B + 12 B, writes 384 B + 2 166 B + 13 B per game and step)
    tracked = gogame.batch_track(states)
    obs = torch.empty_like(states)
    r, ms = event_rate(torch, dev, lambda: gogame.batch_env_step_tracked(tracked, None, rng, 7.5, 'real', True, out=env_out,
                                                                         states_out=obs), count, 32)
    moved = 8 * (5 * N + 1) + 6 * N * N + 25      # tracked board in + out, generator, per-game outputs, the observation
    out['gg_batch_env_step_steps_per_s'] = round(r, 1)
    out['gg_batch_env_step_launch_us'] = round(ms * 1e3, 2)
    out['gg_batch_env_step_bytes_moved_per_step'] = moved
    out['gg_batch_env_step_hbm_frac'] = round(moved * r / 1e9 / HBM_PEAK_GBS, 4)
    out['gg_batch_env_step_x_byte_plane_step_roofline'] = round(algo * r / 1e9 / HBM_PEAK_GBS, 4)
    out['gg_batch_env_step_note'] = ('GoVecEnv.step (layout tracked): gg_batch_env_step_tracked incl. the byte-plane observation; '
                                     'hbm_frac = the bytes this launch really moves (%d B per step: 384 B tracked board in and out, '
                                     'generator, outputs, the 2 166 B observation) against 8 TB/s; x_byte_plane_step_roofline = its '
                                     'rate relative to the roofline of an out-of-place byte-plane step (4 336 B, SURVEY 8d), which '
                                     'it is not bound by' % moved)
    r, _ = event_rate(torch, dev, lambda: gogame.batch_env_step_tracked(tracked, None, rng, 7.5, 'real', True, out=env_out), count, 32)
    out['gg_batch_env_step_no_observation_steps_per_s'] = round(r, 1)
    # the same step (with the observation) as TWO HALF-BATCHES ON TWO STREAMS that run freely, each issuing its next step as
    # soon as its own last one is queued: the head of one half's launch (launch, load, ply - little memory traffic) runs under
    # the write-back of the other's.  What a self-play loop gets that ping-pongs two half-batches between inference and
    # stepping (a single launch that returns the observation of its own step cannot beat 37 - 39 us per 65 536 games)
    try:
        from gymgo_amd.envs import GoVecEnvParts
        halves = GoVecEnvParts(count, N, parts=2, komi=7.5, reward_method='real', device=dev)   # the product's form of it
        for e, (lo, hi) in zip(halves.envs, halves.bounds):
            e.tracked.copy_(tracked[lo:hi]); e.rng.copy_(rng[lo:hi])
        torch.cuda.synchronize(dev)

        def both(n):
            for _ in range(n):
                for h in range(2):
                    halves.step_part(h)
        both(4)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        both(48)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        out['gg_batch_env_step_two_half_batches_two_streams_steps_per_s'] = round(48 * count / dt, 1)
        out['gg_batch_env_step_two_half_batches_us_per_full_step'] = round(dt / 48 * 1e6, 2)
        out['gg_batch_env_step_two_half_batches_note'] = 'GoVecEnvParts(parts=2).step_part: each half on its own stream, free-running'
        del halves
    except Exception as e:
        out['gg_batch_env_step_two_half_batches_note'] = 'failed: %s' % (str(e)[:160],)
    # the headline's launch on boards that STAY in the tracked format (GoVecEnv.rollout, gg_batch_rollout_tracked): no first
    # analysis and no byte-plane write-back per launch - what the byte-plane boundary costs the headline
    tr2, rng2 = tracked.clone(), rng.clone()
    plies = opts['plies_per_step']
    # (VALU-bound, hence clock-bound: the memory-bound extras above leave the shader clock low for the first few launches -
    # measured right behind them this number read 4 % under what the same boards do in tools/exp/ab_layouts.py - so the
    # clock is brought back up with untimed launches first, as the headline's warm-up steps do)
    for _ in range(8):
        gogame.batch_rollout_tracked(tr2, rng2, plies, True)
    r, ms = event_rate(torch, dev, lambda: gogame.batch_rollout_tracked(tr2, rng2, plies, True), count * plies, 8)
    out['fused_rollout_tracked_boards_steps_per_s'] = round(r, 1)
    out['fused_rollout_tracked_boards_launch_ms'] = round(ms, 4)
    del tr2, rng2
    # what the launch boundary (first analysis of the byte planes, write-back, the spread between SIMDs at the end of a launch)
    # costs the headline: the same batch with four times the plies per launch (NOT the headline: `value` stays at --plies-per-step)
    r_long, ms_long = event_rate(torch, dev, lambda: back.rollout(4 * plies, count_steps=False), count * 4 * plies, 3)
    out['fused_rollout_4x_plies_per_launch_steps_per_s'] = round(r_long, 1)
    out['fused_rollout_4x_plies_per_launch_ms'] = round(ms_long, 4)
    # the same step with the move of every game drawn from policy weights (float32 [B, N^2+1]: 1 448 B more to read per
    # game) by the launch itself - what a self-play loop with a policy network runs per ply
    probs = torch.rand((count, N * N + 1), dtype=torch.float32, device=dev)
    r_w, ms_w = event_rate(torch, dev, lambda: gogame.batch_env_step_tracked(tracked, None, rng, 7.5, 'real', True, out=env_out,
                                                                             states_out=obs, weights=probs), count, 32)
    out['gg_batch_env_step_policy_weighted_steps_per_s'] = round(r_w, 1)
    out['gg_batch_env_step_policy_weighted_launch_us'] = round(ms_w * 1e3, 2)
    out['gg_batch_env_step_policy_weighted_vs_uniform'] = round(r_w / out['gg_batch_env_step_steps_per_s'], 4)
    moved_w = moved + 4 * (N * N + 1)          # + the float32 weights of the game
    out['gg_batch_env_step_policy_weighted_bytes_moved_per_step'] = moved_w
    out['gg_batch_env_step_policy_weighted_hbm_frac'] = round(moved_w * r_w / 1e9 / HBM_PEAK_GBS, 4)
    out['gg_batch_env_step_policy_weighted_note'] = ('reads 1.49x the bytes of the uniform-draw step (%d vs %d B per game), but the '
                                                     'extra time is the same for float32 and bfloat16 weights (the tensor handed over '
                                                     'here stays in the 256 MB Infinity Cache): it is the draw itself, ~1 600 VALU '
                                                     'instructions per wave of 16 boards = 1.35 plies (DESIGN 3c)' % (moved_w, moved))
    probs16 = probs.to(torch.bfloat16)      # what a bf16 policy head hands over: widened exactly, half the bytes
    r_h, ms_h = event_rate(torch, dev, lambda: gogame.batch_env_step_tracked(tracked, None, rng, 7.5, 'real', True, out=env_out,
                                                                             states_out=obs, weights=probs16), count, 32)
    out['gg_batch_env_step_policy_weighted_bf16_steps_per_s'] = round(r_h, 1)
    out['gg_batch_env_step_policy_weighted_bf16_launch_us'] = round(ms_h * 1e3, 2)
    out['gg_batch_env_step_policy_weighted_bf16_vs_uniform'] = round(r_h / out['gg_batch_env_step_steps_per_s'], 4)
    out['gg_batch_env_step_policy_weighted_bf16_hbm_frac'] = round((moved + 2 * (N * N + 1)) * r_h / 1e9 / HBM_PEAK_GBS, 4)
    acts_w = torch.empty(count, dtype=torch.int32, device=dev)
    r_s, ms_s = event_rate(torch, dev, lambda: gogame.batch_sample_weighted(states, probs, rng), count, 32)
    out['gg_batch_sample_weighted_boards_per_s'] = round(r_s, 1)
    out['gg_batch_sample_weighted_launch_us'] = round(ms_s * 1e3, 2)
    del tracked, obs, probs, probs16, acts_w
    configs = {}
    F = opts['plies_per_step']
    # --- config 2: 9x9, 4 096 games
    b2 = HipBackend(dev)
    b2.setup(4096, 9, 0)
    for g in range(1, 16):
        b2.rollout(g * 8, g * 256, (g + 1) * 256)
    b2.rollout(F)
    r, ms = event_rate(torch, dev, lambda: b2.rollout(F), 4096 * F, 8)
    r1, ms1 = event_rate(torch, dev, lambda: b2.rollout(1), 4096, 64)
    cus_lib = device_cus()
    configs['config2_9x9_4096_games'] = {
        'fused_rollout_steps_per_s': round(r, 1), 'plies_per_launch': F, 'launch_ms': round(ms, 4),
        'kernel': rollout_kernel_name(9, 4096, F, cus_lib),
        'per_ply_rollout_steps_per_s': round(r1, 1), 'per_ply_launch_us': round(ms1 * 1e3, 2),
        'per_ply_kernel': rollout_kernel_name(9, 4096, 1, cus_lib),
        'per_ply_hbm_frac': round(algo_bytes_per_step(9) * r1 / 1e9 / HBM_PEAK_GBS, 4)}
    # the config's own roofline record, like the headline's: instruction mix and HBM traffic from the committed PMC passes of
    # THIS launch shape (profiles/pmc_rollout.json, tied to the kernel's machine code), rate measured live above
    rl2 = roofline_record(dev, 9, 4096, F, ms, None, opts.get('clocks'))
    rl2['note'] = ('4 096 games of 9x9 are FOUR boards per SIMD: one wave of k_rollout_lat per SIMD (four boards, one row per '
                   'lane), i.e. a latency chain, not an issue-bound kernel - frac is low by construction; what bounds the '
                   'launch is instructions per wave-ply x the ~5 - 7 cycles a lone wave needs per instruction (profiles/r05*)')
    configs['config2_9x9_4096_games']['roofline'] = rl2
    # the same one-ply launches as a hipGraph of 64 (captured once, replayed): at this batch size a launch through the
    # Python API is paced by the host (ctypes call + hipLaunchKernel per ply), not by the 4 096 boards - a loop that steps
    # small batches ply by ply should replay a graph (every entry point is capturable: no allocation, no synchronisation)
    try:
        side = torch.cuda.Stream(device=dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            b2.rollout(1, count_steps=False)
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(64):
                b2.rollout(1, count_steps=False)
        rg, msg = event_rate(torch, dev, graph.replay, 4096 * 64, 16)
        configs['config2_9x9_4096_games'].update({
            'per_ply_graph_steps_per_s': round(rg, 1), 'per_ply_graph_launch_us': round(msg * 1e3 / 64, 2),
            'per_ply_graph_hbm_frac': round(algo_bytes_per_step(9) * rg / 1e9 / HBM_PEAK_GBS, 4),
            'per_ply_graph_note': '64 one-ply launches of gg_batch_rollout captured in one hipGraph, replayed 16 times'})
        del graph
    except Exception as e:      # no graph support on this stack: the API-loop number above stands alone
        configs['config2_9x9_4096_games']['per_ply_graph_note'] = 'hipGraph capture failed: %s' % (str(e)[:160],)
    # the same games kept in the TRACKED layout (GoVecEnv's default: the liberty classes travel with the board, no first
    # analysis per launch) - gg_batch_rollout_tracked: fused, one ply per launch through the API, and as a hipGraph of 64
    try:
        tr2 = gogame.batch_track(b2.states)
        rg2 = b2.rng.clone()
        for _ in range(3):
            gogame.batch_rollout_tracked(tr2, rg2, F, True)
        rt, mst = event_rate(torch, dev, lambda: gogame.batch_rollout_tracked(tr2, rg2, F, True), 4096 * F, 8)
        rt1, mst1 = event_rate(torch, dev, lambda: gogame.batch_rollout_tracked(tr2, rg2, 1, True), 4096, 64)
        rec_t = {'layout': 'tracked boards (uint32 [B][5N+1]), gg_batch_rollout_tracked', 'kernel': 'k_rollout_lat<9, true, true, 2>',
                 'fused_rollout_steps_per_s': round(rt, 1), 'launch_ms': round(mst, 4),
                 'per_ply_rollout_steps_per_s': round(rt1, 1), 'per_ply_launch_us': round(mst1 * 1e3, 2)}
        side = torch.cuda.Stream(device=dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            gogame.batch_rollout_tracked(tr2, rg2, 1, True)
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(64):
                gogame.batch_rollout_tracked(tr2, rg2, 1, True)
        rg_, msg_ = event_rate(torch, dev, graph.replay, 4096 * 64, 16)
        rec_t.update({'per_ply_graph_steps_per_s': round(rg_, 1), 'per_ply_graph_launch_us': round(msg_ * 1e3 / 64, 2)})
        del graph
        configs['config2_9x9_4096_games']['tracked_layout'] = rec_t
    except Exception as e:
        configs['config2_9x9_4096_games']['tracked_layout'] = {'note': 'failed: %s' % (str(e)[:160],)}
    del b2
    # --- config 4's per-GPU batch (131 072 games) on this one GPU: the base for weak-scaling ratios
    if opts['world'] == 1 and N == 19 and count != 131072:
        b4 = HipBackend(dev)
        b4.setup(131072, N, 0)
        for g in range(1, 16):
            b4.rollout(g * opts['desync'] // 16 if opts['desync'] else 0, g * 8192, (g + 1) * 8192)
        b4.rollout(F)
        r, ms = event_rate(torch, dev, lambda: b4.rollout(F), 131072 * F, 4)
        configs['config4_per_gpu_batch_131072_games_on_one_gpu'] = {'fused_rollout_steps_per_s': round(r, 1),
                                                                   'plies_per_launch': F, 'launch_ms': round(ms, 4)}
        del b4
    # --- config 5: 8 192 mid-game parents, padded 362-slot expansion (+ the un-padded form of the same parents)
    if N == 19 and count >= 8192:
        configs['config5_children_8192_parents'] = children_record(
            torch, dev, states[:8192], 'the first 8 192 games of the resident batch (stationary mix: every game phase)', by_phase=True)
    # --- config 1: one 7x7 game through GoEnv.step (device round trip per step: plumbing, not a throughput path)
    import numpy as np
    env = make('gym_go:go-v0', size=7)
    env.reset()
    rs = np.random.default_rng(1)
    n_steps, t0 = 0, time.perf_counter()
    while n_steps < 600:
        if env.done:
            env.reset()
        valid = np.flatnonzero(env.valid_moves())
        env.step(int(valid[rs.integers(len(valid))]))
        n_steps += 1
    configs['config1_7x7_single_game_GoEnv_step'] = {
        'steps_per_s': round(n_steps / (time.perf_counter() - t0), 1),
        'note': 'GoEnv.step through the reference-style API (NumPy in / out) driven by a uniform-random policy on the host: one '
                'launch (gg_batch_env_step_scored) on a record in pinned, device-mapped host memory + a stream wait per step, no '
                'copy in either direction; latency-bound plumbing by construction'}
    out['configs'] = configs
    # --- what this box streams ("also report against a measured device-copy bandwidth on the box")
    big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(big)
    r_fill, _ = event_rate(torch, dev, lambda: big.zero_(), float(1 << 30), 5)
    r_copy, _ = event_rate(torch, dev, lambda: dst.copy_(big), float(2 << 30), 5)
    del big, dst
    out['box_bandwidth'] = {'fill_GBps': round(r_fill / 1e9, 1), 'copy_read_plus_write_GBps': round(r_copy / 1e9, 1),
                            'note': '1 GiB torch fill / copy on this box; the per-ply fractions above are against the 8 TB/s vendor peak',
                            # next_states reads a board and writes a board, like a copy; the tracked env step really
                            # moves 5 (5N+1) + 12 B in and the same + the 6 N^2 B observation + 13 B out, 7/8 of it writes
                            'gg_batch_next_states_frac_of_copy': round(algo * out['gg_batch_next_states_steps_per_s'] / r_copy, 4),
                            'gg_batch_env_step_bytes_moved_per_step': 8 * (5 * N + 1) + 6 * N * N + 25,
                            'gg_batch_env_step_frac_of_fill': round((8 * (5 * N + 1) + 6 * N * N + 25)
                                                                    * out['gg_batch_env_step_steps_per_s'] / r_fill, 4)}
    # --- throughput against batch size (the dispatch thresholds are one box's tuning - show what they
    # give): 19x19 and 9x9, fused (F plies per launch) and one ply per launch, with the kernel that served each shape
    sweep = {}
    for n_s, sizes in ((19, (1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072)), (9, (1024, 4096, 16384, 65536))):
        rows = []
        for games in sizes:
            bs = HipBackend(dev)
            bs.setup(games, n_s, 0)
            chunk = max(1, games // 16)
            for g in range(1, 16):
                bs.rollout(g * (40 if n_s == 19 else 8), g * chunk, min(games, (g + 1) * chunk))
            t_up = time.perf_counter()          # (the launches after an idle gap run slow - clock ramp: 40 ms of the shape first)
            while time.perf_counter() - t_up < 0.04:
                bs.rollout(F, count_steps=False)
                bs.sync()
            rf_, msf = event_rate(torch, dev, lambda: bs.rollout(F, count_steps=False), games * F, 4 if games >= 32768 else 8)
            r1_, ms1_ = event_rate(torch, dev, lambda: bs.rollout(1, count_steps=False), games, 24)
            rows.append({'games': games, 'fused_steps_per_s': round(rf_, 1), 'fused_launch_ms': round(msf, 4),
                         'fused_kernel': rollout_kernel_name(n_s, games, F, cus_lib),
                         'one_ply_steps_per_s': round(r1_, 1), 'one_ply_launch_us': round(ms1_ * 1e3, 2),
                         'one_ply_kernel': rollout_kernel_name(n_s, games, 1, cus_lib)})
            del bs
        sweep['%dx%d' % (n_s, n_s)] = rows
    out['batch_sweep'] = {'plies_per_fused_launch': F, 'sizes': sweep,
                          'note': 'gg_batch_rollout on byte planes through the Python API, HIP events over back-to-back launches; '
                                  'stationary mix (slices de-synchronised); the kernel column mirrors gg_kernels.hip for %d CUs' % cus_lib}
    out['note'] = ('per-ply rates: HIP events over back-to-back calls through the Python API on the resident config-3 batch; '
                   'configs: one driver-timed number per BASELINE config that is not the headline')
    return out, per_ply


def flat_config_scalars(line, also, res):
    """One flat scalar per BASELINE config and per protocol fact, INSIDE `roofline` (the driver's record keeps `roofline`,
    `config` and `cpu_baseline` whole and drops every other nested dict of the line - round 5's config 1 / 2 / 5 numbers and
    the clock only survived in builder-run profiles/).  Every figure is measured in this run; None = that part did not run."""
    def dig(d, *path):
        for k in path:
            d = d.get(k) if isinstance(d, dict) else None
        return d
    clocks = res.get('clocks') or {}
    cfg = (also or {}).get('configs') or {}
    c2, c5, c1 = cfg.get('config2_9x9_4096_games'), cfg.get('config5_children_8192_parents'), cfg.get('config1_7x7_single_game_GoEnv_step')
    c4 = cfg.get('config4_per_gpu_batch_131072_games_on_one_gpu')
    flat = {
        'sclk_mhz': clocks.get('sclk_mhz'),
        'settle_launches': clocks.get('launches'), 'settle_seconds': clocks.get('seconds'),
        'settle_first_window_launch_ms': clocks.get('first_window_launch_ms'),
        'settle_last_window_launch_ms': clocks.get('last_window_launch_ms'),
        'timed_launch_ms': line['roofline'].get('launch_ms'),
        'per_ply_frac': dig(line['roofline'], 'per_ply', 'frac'),
        'per_ply_steps_per_s': dig(line['roofline'], 'per_ply', 'env_steps_per_s'),
        'config1_steps_per_s': dig(c1, 'steps_per_s'),
        'config2_fused_steps_per_s': dig(c2, 'fused_rollout_steps_per_s'), 'config2_fused_launch_ms': dig(c2, 'launch_ms'),
        'config2_frac': dig(c2, 'roofline', 'frac'),
        'config2_per_ply_us': dig(c2, 'per_ply_graph_launch_us'), 'config2_per_ply_steps_per_s': dig(c2, 'per_ply_graph_steps_per_s'),
        'config2_per_ply_hbm_frac': dig(c2, 'per_ply_graph_hbm_frac'),
        'config2_tracked_fused_steps_per_s': dig(c2, 'tracked_layout', 'fused_rollout_steps_per_s'),
        'config2_tracked_per_ply_us': dig(c2, 'tracked_layout', 'per_ply_graph_launch_us'),
        'config4_per_gpu_batch_steps_per_s': dig(c4, 'fused_rollout_steps_per_s'),
        'config5_parents_per_s': dig(c5, 'parents_per_s'), 'config5_launch_ms': dig(c5, 'launch_ms'),
        'config5_frac': dig(c5, 'roofline', 'frac'),
        'config5_compact_parents_per_s': dig(c5, 'compact', 'parents_per_s'), 'config5_compact_frac': dig(c5, 'compact', 'frac'),
        'env_step_tracked_steps_per_s': (also or {}).get('gg_batch_env_step_steps_per_s'),
        'env_step_policy_weighted_vs_uniform': (also or {}).get('gg_batch_env_step_policy_weighted_vs_uniform'),
    }
    # the same launch shape elsewhere in this run (the batch sweep times it seconds later, after other work): round 5's
    # driver run had the two 7 % apart; with the settle phase they must agree
    for row in dig(also, 'batch_sweep', 'sizes', '%dx%d' % (line['config']['board'], line['config']['board'])) or []:
        if row.get('games') == line['config']['games_per_gpu'] and line['n_gpus'] == 1:
            flat['sweep_same_shape_launch_ms'] = row.get('fused_launch_ms')
            if flat['timed_launch_ms']:
                flat['timed_vs_sweep'] = round(flat['timed_launch_ms'] / row['fused_launch_ms'], 4)
        if row.get('games') == 4096:
            flat['mid_batch_4096_games_fused_steps_per_s'] = row.get('fused_steps_per_s')
            flat['mid_batch_4096_games_fused_launch_ms'] = row.get('fused_launch_ms')
    return flat


# ------------------------------------------------------------------------------------------------ entry
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20, help='timed launches (each --plies-per-step plies for every game)')
    ap.add_argument('--warmup', type=int, default=5, help='untimed launches before the timed ones')
    ap.add_argument('--size', type=int, default=19)
    ap.add_argument('--games-per-gpu', type=int, default=0, help='0 = 65536 at 1 GPU, 131072 per GPU otherwise')
    ap.add_argument('--plies-per-step', '--fuse', dest='plies_per_step', type=int,
                    default=256, help='plies per kernel launch (= per bench step)')
    ap.add_argument('--burn-in', type=int, default=1, help='untimed launches before warm-up (same shape as the timed ones)')
    ap.add_argument('--desync', type=int, default=640, help='spread of extra burn-in plies across the batch (0 = lock-step)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=6.0, help='wall seconds per CPU-baseline worker')
    ap.add_argument('--cpu-workers', type=int, default=0, help='CPU-baseline worker processes (0 = every usable core)')
    ap.add_argument('--no-also', action='store_true', help='skip the untimed extras (clean profiling passes)')
    ap.add_argument('--comm', choices=('nccl', 'gloo'), default='nccl',
                    help='process-group backend for barrier + reductions (gloo: rehearse the N > 1 path on a box with fewer '
                         'GPUs than ranks - ranks then share devices round-robin)')
    return ap.parse_args(argv)


def _flush_c_stdio():
    """RCCL prints a version banner through C stdio (buffered when stdout is a pipe): push it out now, so that the JSON
    line printed at the very end is the LAST line of stdout."""
    try:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _spawned(local_rank, world, port, argv):
    os.environ.update({'RANK': str(local_rank), 'LOCAL_RANK': str(local_rank), 'WORLD_SIZE': str(world),
                       'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'HSA_ENABLE_IPC_MODE_LEGACY': '0'})
    main(argv)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher - one process per GPU, rendezvous on 127.0.0.1
        import torch.multiprocessing as mp
        mp.spawn(_spawned, args=(args.gpus, _free_port(), list(argv)), nprocs=args.gpus, join=True)
        return
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    args.gpus = world

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        # every world size carries the CPU path of the same box in the same run (north_star); on rank 0, before the GPU
        # context and the process group exist (spawned workers) - the other ranks wait for it at the rendezvous
        cpu = cpu_baseline(args.size, args.cpu_seconds, args.cpu_workers or None)

    import torch
    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit('bench.py: no GPU visible (the HIP path has no CPU fallback)')
    if world > ndev and args.comm == 'nccl':
        # RCCL refuses two ranks on one device with an opaque "duplicate GPU" error deep inside init: say it here
        raise SystemExit('bench.py: %d ranks but %d visible GPU(s): --comm nccl needs one GPU per rank (use --comm gloo to '
                         'rehearse the N > 1 path with ranks sharing devices round-robin)' % (world, ndev))
    local_rank %= ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    numa = pin_to_gpu_numa_node(torch, dev) if world > 1 else None      # one rank per GPU: each on its GPU's NUMA node
    # under a launcher (WORLD_SIZE set, also with one rank) the process group is real: barrier and reductions run over RCCL
    use_dist = 'WORLD_SIZE' in os.environ
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('NCCL_DEBUG', 'WARN')       # no version banner on stdout unless asked for
        # rank 0 joins after its CPU baseline (cores x --cpu-seconds of wall time + worker start-up): the others wait that long
        import datetime
        patience = datetime.timedelta(seconds=600 + 4 * int(args.cpu_seconds))
        if args.comm == 'nccl':
            dist.init_process_group('nccl', device_id=dev, timeout=patience)
        else:
            dist.init_process_group('gloo', timeout=patience)
        dist.barrier()                                    # the communicator exists from here on (banner printed, if any)
        _flush_c_stdio()

    opts = {'size': args.size, 'plies_per_step': max(1, args.plies_per_step), 'steps': max(1, args.steps),
            'warmup': max(0, args.warmup), 'games_per_gpu': args.games_per_gpu or (65536 if world == 1 else 131072),
            'desync': args.desync, 'burn_in_steps': max(0, args.burn_in), 'world': world}
    back = HipBackend(dev)
    res = run_rank(rank, world, back, opts, dist if use_dist else None)

    # N > 1: the weak-scaling base of THIS run - rank 0 repeats the timed launches alone (every other rank idle at the barrier:
    # its shard is the per-GPU batch, the launch shape is the timed one) - so that the line carries its own efficiency =
    # value(N) / (N x base) instead of leaving the division to a second run
    solo = None
    if world > 1:
        if use_dist:
            dist.barrier()
        if rank == 0:
            import time as _t
            back.sync()
            t0 = _t.perf_counter()
            for _ in range(opts['steps']):
                back.rollout(opts['plies_per_step'], count_steps=False)
            back.sync()
            solo = opts['steps'] * opts['plies_per_step'] * back.count / (_t.perf_counter() - t0)
        if use_dist:
            dist.barrier()
    if rank == 0:
        N, F, K, W = opts['size'], opts['plies_per_step'], opts['steps'], opts['warmup']
        opts['clocks'] = res.get('clocks')
        also, per_ply = ({}, None) if args.no_also else extras(dev, back, opts)
        launch_ms = res['kernel_ms'] / K
        line = {
            'metric': METRIC, 'value': round(res['value'], 1), 'unit': 'env steps/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': round(res['wall_s'] * 1e3 / K, 6), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
            'config': {
                'workload': '%dx%d, %d parallel games%s, uniform-random rollouts with auto-reset; one bench step = one '
                            'launch of %d plies for every game' % (N, N, res['total_games'],
                                                                   '' if world == 1 else ' (%d per GPU)' % res['games_per_gpu'], F),
                'board': N, 'games': res['total_games'], 'games_per_gpu': res['games_per_gpu'], 'plies_per_step': F,
                'env_steps_per_bench_step': F * res['total_games'], 'burn_in_steps': opts['burn_in_steps'],
                'desync_plies': opts['desync'], 'sharding': 'batch split across ranks by global game index, no collective',
            },
            'roofline': roofline_record(dev, N, res['count'], F, launch_ms, per_ply, res.get('clocks')),
            'clocks': res.get('clocks'),
            # what the communicator saw (None at a plain one-process run): backend, its world size, the ranks an
            # all-reduce of ones counted; and every rank's own average launch time (HIP events on its stream)
            'comm': res['comm'], 'rccl_world_size': res['comm']['world_size'] if res['comm']['backend'] else None,
            'per_rank_launch_ms': [round(x, 5) for x in res['per_rank_launch_ms']],
            # every rank about itself: first game of its shard, steps it played, device (index + identity digest), clock
            'per_rank': res.get('per_rank'), 'distinct_devices': res.get('distinct_devices'),
            'per_gpu_steps_per_s': [round(F * res['games_per_gpu'] / (x * 1e-3), 1) for x in res['per_rank_launch_ms']],
        }
        line['roofline'].update(flat_config_scalars(line, also, res))
        if numa is not None:
            line['rank0_numa'] = numa
        if solo:
            line['weak_scaling_base'] = {'games': res['games_per_gpu'], 'steps_per_s_one_gpu': round(solo, 1),
                                         'how': 'rank 0 alone, the other ranks idle at a barrier: the same %d launches of the same shape, right after the timed region' % K}
            line['efficiency'] = round(res['value'] / (world * solo), 4)
        if cpu is not None:
            line['cpu_baseline'] = cpu
        base = (also.get('configs') or {}).get('config4_per_gpu_batch_131072_games_on_one_gpu') if also else None
        if world == 1 and base:
            # weak-scaling base: the per-GPU batch of the N > 1 lines (131 072 games) on ONE GPU, same launch shape -
            # efficiency at N GPUs = value(N) / (N x this)
            line['weak_scaling_base'] = {'games': 131072, 'steps_per_s_one_gpu': base['fused_rollout_steps_per_s'],
                                         'launch_ms': base['launch_ms']}
        elif world == 1 and res['games_per_gpu'] == 131072:
            line['weak_scaling_base'] = {'games': 131072, 'steps_per_s_one_gpu': round(res['value'], 1),
                                         'launch_ms': round(launch_ms, 5)}
        line['also'] = also
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    _flush_c_stdio()
    if rank == 0:
        print(json.dumps(line), flush=True)   # the last line of stdout


if __name__ == '__main__':
    main()
