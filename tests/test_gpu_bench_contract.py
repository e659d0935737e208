"""-m gpu: bench.py's driver-facing contract, run the way the driver runs it (a subprocess; plain for one GPU, under
torch.distributed.run and self-spawned for two ranks - rehearsed over gloo with both ranks on this box's GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline')


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(cmd, timeout=420):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    return json.loads(lines[-1])          # the JSON line is the LAST line of stdout


def _check(line, n_gpus, steps, warmup):
    for k in REQUIRED:
        assert k in line, k
    assert line['n_gpus'] == n_gpus and line['steps'] == steps and line['warmup'] == warmup
    assert line['metric'].startswith('env steps/sec')
    assert line['higher_is_better'] is True and line['scaling'] == 'weak' and line['vs_baseline'] is None
    assert line['dtype'] == 'u8' and line['data'] == 'synthetic' and 'workload' in line['config']
    assert line['value'] > 1e8 and line['ms_per_step'] > 0
    # value is the whole job: every rank's games x plies per step / the step time
    per_step = line['config']['env_steps_per_bench_step']
    assert per_step == line['config']['games'] * line['config']['plies_per_step']
    assert line['value'] == pytest.approx(per_step / (line['ms_per_step'] * 1e-3), rel=1e-3)
    roof = line['roofline']
    assert roof['bound'] == 'valu' and roof['peak'] > 0
    if roof['frac'] is not None:
        assert 0 < roof['frac'] <= 1 and 0 < roof['hbm']['frac'] <= 1
        assert roof['traffic'] is not None and roof['traffic'] > 0


def test_bench_one_gpu_plain_invocation():
    line = _run([sys.executable, 'bench.py', '--gpus', '1', '--steps', '3', '--warmup', '1', '--cpu-seconds', '0.5',
                 '--cpu-workers', '8'])
    _check(line, 1, 3, 1)
    assert line['comm']['backend'] is None and line['rccl_world_size'] is None and len(line['per_rank_launch_ms']) == 1
    assert line['weak_scaling_base']['games'] == 131072 and line['weak_scaling_base']['steps_per_s_one_gpu'] > 1e8
    assert 'clocks' in line and 'measured_clock' in line['roofline']
    mc, ck = line['roofline']['measured_clock'], line['clocks']
    if ck and ck.get('sclk_mhz'):      # ONE clock: the settled figure sampled before the timed region prices the fraction
        assert mc and abs(mc['sclk_mhz'] - ck['sclk_mhz']) <= 0.02 * ck['sclk_mhz'] and ck['seconds'] >= 0.3
        assert ck['launches'] >= 24 and ck['first_window_launch_ms'] > 0 and ck['last_window_launch_ms'] > 0      # the adaptive settle phase (round 6)
    assert line['config']['games'] == 65536 and line['config']['board'] == 19 and line['config']['plies_per_step'] == 256
    assert line['roofline']['frac'] is not None                       # the committed PMC record matches the default shape
    assert 0 < line['roofline']['per_ply']['frac'] <= 1
    cpu = line['cpu_baseline']
    assert cpu['kind'] == 'port' and cpu['cores'] == 8 and cpu['value'] > 0 and cpu['unit'] == line['unit']
    ratio = cpu['port_vs_reference_speed']    # from oracle/ref_harness/speed_calibration.json, with its range
    assert 1 < ratio['min'] <= ratio['mean'] <= ratio['max'] and ratio['repetitions'] >= 3 and ratio['positions'] >= 2000
    est = cpu['reference_estimate_steps_per_s']
    assert est['range'][0] <= est['mean'] <= est['range'][1] < cpu['value']
    also = line['also']
    assert also['gg_batch_env_step_hbm_frac'] < also['gg_batch_env_step_x_byte_plane_step_roofline'] <= 1.2
    assert also['gg_batch_env_step_steps_per_s'] > 0 and set(also['configs']) >= {
        'config2_9x9_4096_games', 'config5_children_8192_parents', 'config1_7x7_single_game_GoEnv_step'}
    c2 = also['configs']['config2_9x9_4096_games']
    assert c2['kernel'].startswith('k_rollout_lat<9') and c2['roofline']['kernel'] == c2['kernel']
    assert c2['roofline']['frac'] is not None and c2['roofline']['pmc_stale'] is False     # the committed PMC pass is of this code
    sweep = also['batch_sweep']['sizes']
    assert [r['games'] for r in sweep['19x19']] == [1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072]
    assert all(r['fused_steps_per_s'] > 0 and r['one_ply_steps_per_s'] > 0 for rows in sweep.values() for r in rows)


@pytest.mark.parametrize('launcher', ['torch.distributed.run', 'self-spawn'])
def test_bench_two_ranks(launcher):
    tail = ['bench.py', '--gpus', '2', '--comm', 'gloo', '--games-per-gpu', '16384', '--steps', '2', '--warmup', '1', '--no-also',
            '--cpu-seconds', '0.5', '--cpu-workers', '4']
    if launcher == 'self-spawn':
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', str(_free_port())] + tail
    line = _run(cmd)
    _check(line, 2, 2, 1)
    assert line['config']['games'] == 32768 and line['config']['games_per_gpu'] == 16384
    # the N > 1 line is complete and self-evidencing: the CPU path of the same box in the same run (rank 0), the world
    # size as the communicator counted it (an all-reduce of ones), every rank's own launch time
    assert line['cpu_baseline']['kind'] == 'port' and line['cpu_baseline']['cores'] == 4 and line['cpu_baseline']['value'] > 0
    assert line['comm'] == {'backend': 'gloo', 'world_size': 2, 'ranks_counted': 2} and line['rccl_world_size'] == 2
    assert len(line['per_rank_launch_ms']) == 2 and all(x > 0 for x in line['per_rank_launch_ms'])
    assert len(line['per_gpu_steps_per_s']) == 2
    assert line['value'] <= sum(line['per_gpu_steps_per_s']) * 1.001    # the whole job cannot beat the sum of its ranks
    # the N > 1 line carries its own weak-scaling base (rank 0 alone, same shape) and the efficiency against it; the ranks are
    # pinned to their GPU's NUMA node where the platform says which that is
    assert line['weak_scaling_base']['games'] == 16384 and line['weak_scaling_base']['steps_per_s_one_gpu'] > 0
    assert line['efficiency'] == pytest.approx(line['value'] / (2 * line['weak_scaling_base']['steps_per_s_one_gpu']), rel=1e-3)
    assert 0 < line['efficiency'] <= 1.05 and 'numa_node' in line['rank0_numa']          # (two ranks share ONE GPU here: ~0.5)


def test_bench_one_rank_over_rccl():
    """RCCL itself on this box: bench.py under torch.distributed.run with ONE rank and --comm nccl - init_process_group('nccl'),
    the barriers and the all-reduces of the N > 1 path run over RCCL (world size 1), so the first contact with 8 ranks is not
    the first contact with the communicator."""
    line = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                 '--master-port', str(_free_port()), 'bench.py', '--gpus', '1', '--comm', 'nccl', '--steps', '2', '--warmup', '1',
                 '--no-also', '--no-cpu-baseline'])
    _check(line, 1, 2, 1)
    assert line['comm'] == {'backend': 'nccl', 'world_size': 1, 'ranks_counted': 1} and line['rccl_world_size'] == 1
    assert len(line['per_rank']) == 1 and line['per_rank'][0]['first_game'] == 0


def test_bench_per_rank_records_and_nccl_device_check():
    """The N > 1 line says what every rank did - first game of its shard, steps played, device (two ranks rehearsed on ONE
    GPU over gloo show up as equal device identities: `distinct_devices` 1) - and `--comm nccl` with more ranks than
    visible GPUs is refused with a clear message instead of RCCL's duplicate-GPU failure."""
    import torch
    line = _run([sys.executable, 'bench.py', '--gpus', '2', '--comm', 'gloo', '--games-per-gpu', '16384', '--steps', '2', '--warmup', '1',
                 '--no-also', '--no-cpu-baseline'])
    per = line['per_rank']
    assert [p['first_game'] for p in per] == [0, 16384] and [p['steps_played'] for p in per] == [2 * 256 * 16384] * 2
    ndev = torch.cuda.device_count()
    assert line['distinct_devices'] == min(2, ndev)
    assert line['roofline']['pmc_stale'] in (True, False) and 'kernel_code_sha16' in line['roofline']
    if ndev < 2:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
            env.pop(k, None)
        res = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--comm', 'nccl', '--steps', '1', '--warmup', '0', '--no-also',
                              '--no-cpu-baseline'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert res.returncode != 0 and 'one GPU per rank' in (res.stderr + res.stdout)
