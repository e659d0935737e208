#!/usr/bin/env python
"""Condense gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) into tracked files under
profiles/: <tag>_bench.json, <tag>_kernel_stats.csv, <tag>_summary.md and hbm_traffic.json (read by bench.py)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src = os.path.join(ROOT, 'gpurun_out', tag)
dst = os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)

bench = json.loads(open(os.path.join(src, 'bench.json')).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(dst, tag + '_bench.json'), 'w'), indent=1)
F = bench['config']['plies_per_launch']
games = bench['config']['games_per_gpu']
shutil.copy(os.path.join(src, 'kt', 'kt_kernel_stats.csv'), os.path.join(dst, tag + '_kernel_stats.csv'))
stats = list(csv.DictReader(open(os.path.join(src, 'kt', 'kt_kernel_stats.csv'))))


def counters(sub, want='rollout'):
    """Per-dispatch counters of the full-batch launches (the de-synchronising burn-in launches run on 1/16 slices)."""
    rows = list(csv.DictReader(open(os.path.join(src, sub, 'p_counter_collection.csv'))))
    full = max(int(r['Grid_Size']) for r in rows if want in r['Kernel_Name'])
    per = collections.defaultdict(dict)
    for r in rows:
        if want in r['Kernel_Name'] and int(r['Grid_Size']) == full:
            d = per[int(r['Dispatch_Id'])]
            d[r['Counter_Name']] = float(r['Counter_Value'])
            d['_vgpr'], d['_sgpr'], d['_lds'] = r.get('VGPR_Count'), r.get('SGPR_Count'), r.get('LDS_Block_Size')
    ids = sorted(per)
    return [per[i] for i in ids[2:]]   # skip the first two (cold) launches; every launch is F plies


md = ['# %s profile summary (MI355X, `bench.py --fuse %d`, %d games of %dx%d)\n' % (
    tag, F, games, bench['config']['board'], bench['config']['board'])]
md.append('bench line: value = %.4g %s, ms_per_step = %.5f, roofline.frac = %.4f (achieved %.1f GB/s algorithmic)\n'
          % (bench['value'], bench['unit'], bench['ms_per_step'], bench['roofline']['frac'], bench['roofline']['achieved']))
md.append('## rocprofv3 --kernel-trace --stats (same command)\n')
md.append('| kernel | calls | avg ns | total % |\n|---|---|---|---|')
for r in stats[:4]:
    md.append('| `%s` | %s | %.0f | %s |' % (r['Name'][:70], r['Calls'], float(r['AverageNs']), r['Percentage']))
trace = [r for r in csv.DictReader(open(os.path.join(src, 'kt', 'kt_kernel_trace.csv'))) if 'k_rollout' in r['Kernel_Name']]
full = max(int(r['Grid_Size_X']) for r in trace)
fullrows = [r for r in trace if int(r['Grid_Size_X']) == full]
kname = fullrows[0]['Kernel_Name']
durs = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in fullrows]
other = [r for r in trace if int(r['Grid_Size_X']) != full]
md.append('\nbench.py live launch_ms = %.4f; rocprofv3 average of the %d full-batch launches of `%s` (burn-in, warm-up, '
          'timed: %d plies each) = %.4f ms.  The other %d rollout launches of the trace are the de-synchronising burn-in '
          'launches on 1/16 slices of the batch (4 096 games each: below 8 192 games `gg_batch_rollout` uses the v2 kernel).'
          % (bench['roofline']['launch_ms'], len(durs), kname.split('(')[0].replace('void gg::', ''), F,
             sum(durs) / len(durs) / 1e6, len(other)))

steps = games * F
traffic = {}
for name, sub in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write')):
    vals = [d[name] for d in counters(sub)]
    traffic[name] = sum(vals) / len(vals)
fetch_b = traffic['FETCH_SIZE'] * 1024 * 2   # gfx950: FETCH_SIZE reports 1/2 of a wide coalesced stream (MI355X_MICROARCH.md, HBM)
write_b = traffic['WRITE_SIZE'] * 1024
md.append('\n## HBM traffic per launch (PMC, separate passes; FETCH_SIZE x2 gfx950 correction, KiB units)\n')
md.append('FETCH_SIZE = %.0f KiB -> %.1f MB read; WRITE_SIZE = %.0f KiB -> %.1f MB written; total %.1f MB per launch '
          '= %.0f B per game per launch (algorithmic: %d B x %d steps = %.1f MB)'
          % (traffic['FETCH_SIZE'], fetch_b / 1e6, traffic['WRITE_SIZE'], write_b / 1e6, (fetch_b + write_b) / 1e6,
             (fetch_b + write_b) / games, bench['roofline']['algorithmic_bytes_per_step'], steps,
             bench['roofline']['algorithmic_bytes_per_step'] * steps / 1e6))
json.dump({'size': bench['config']['board'], 'fuse': F, 'games': games,
           'bytes_per_launch': round(fetch_b + write_b), 'fetch_bytes': round(fetch_b), 'write_bytes': round(write_b),
           'source': 'profiles/%s_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)' % tag},
          open(os.path.join(dst, 'hbm_traffic.json'), 'w'), indent=1)

md.append('\n## instruction mix per env step (PMC / (games x plies))\n')
for sub in ('pmc_inst', 'pmc_act'):
    c = counters(sub)
    keys = [k for k in c[0] if not k.startswith('_')]
    avg = {k: sum(d[k] for d in c) / len(c) / steps for k in keys}
    md.append('- ' + ', '.join('%s %.1f' % (k, v) for k, v in sorted(avg.items())))
    md.append('  (LDS %s B per 64-thread workgroup)' % c[0]['_lds'])
md.append('\nSQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* are quad-cycles. The kernel is VALU-issue / dependency-latency bound, '
          'not HBM-bound: per-op issue costs (2 or 4 cycles per wave64) are in profiles/r01_ubench_valu_rates.txt.')
ops_stats = os.path.join(src, 'kt_ops', 'kt_kernel_stats.csv')
if os.path.exists(ops_stats):
    shutil.copy(ops_stats, os.path.join(dst, tag + '_ops_kernel_stats.csv'))
    shutil.copy(os.path.join(src, 'ops.json'), os.path.join(dst, tag + '_ops.json'))
    md.append('\n## the other entry points (`tools/bench_ops.py`, rocprofv3 --kernel-trace --stats; %s_ops.json has the rates)\n' % tag)
    md.append('| kernel | calls | avg ns |\n|---|---|---|')
    for r in list(csv.DictReader(open(ops_stats)))[:14]:
        md.append('| `%s` | %s | %.0f |' % (r['Name'][:80], r['Calls'], float(r['AverageNs'])))
open(os.path.join(dst, tag + '_summary.md'), 'w').write('\n'.join(md) + '\n')
print('\n'.join(md))
