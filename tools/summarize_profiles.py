#!/usr/bin/env python
"""Condense gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) into tracked files under profiles/:
<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_summary.md, <tag>_ops*.{json,csv} and pmc_rollout.json - the
instruction mix and HBM traffic per launch shape that bench.py's roofline record reads."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
src = os.path.join(ROOT, 'gpurun_out', tag)
dst = os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)

bench = json.loads(open(os.path.join(src, 'bench.json')).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(dst, tag + '_bench.json'), 'w'), indent=1)
F = bench['config']['plies_per_step']
games = bench['config']['games_per_gpu']
N = bench['config']['board']
kernel = bench['roofline']['kernel']
shutil.copy(os.path.join(src, 'kt', 'kt_kernel_stats.csv'), os.path.join(dst, tag + '_kernel_stats.csv'))
stats = list(csv.DictReader(open(os.path.join(src, 'kt', 'kt_kernel_stats.csv'))))


def counter_rows(sub):
    return list(csv.DictReader(open(os.path.join(src, sub, 'p_counter_collection.csv'))))


def counters(sub, want, skip=2):
    """Per-dispatch counters of the full-batch launches of kernel `want` (the de-synchronising burn-in launches run on
    1/16 slices: smaller grids), without the first `skip` (cold) ones."""
    rows = [r for r in counter_rows(sub) if want in r['Kernel_Name']]
    full = max(int(r['Grid_Size']) for r in rows)
    per = collections.defaultdict(dict)
    for r in rows:
        if int(r['Grid_Size']) == full:
            d = per[int(r['Dispatch_Id'])]
            d[r['Counter_Name']] = float(r['Counter_Value'])
            d['_vgpr'], d['_sgpr'], d['_lds'] = r.get('VGPR_Count'), r.get('SGPR_Count'), r.get('LDS_Block_Size')
    ids = sorted(per)
    return [per[i] for i in ids[skip:]]


def mean(xs):
    xs = list(xs)
    return sum(xs) / len(xs)


kshort = kernel.split('<')[0]
md = ['# %s profile summary (MI355X, `bench.py --steps 20 --warmup 5 --plies-per-step %d`, %d games of %dx%d)\n' % (tag, F, games, N, N)]
rf = bench['roofline']
md.append('bench line: value = %.4g %s, ms_per_step = %.5f (one step = one launch of %d plies); roofline: bound %s, '
          'achieved %s %s of peak %s -> frac %s; fused HBM frac %s; per-ply `%s` HBM frac %s\n'
          % (bench['value'], bench['unit'], bench['ms_per_step'], F, rf['bound'], rf.get('achieved'), rf['unit'], rf['peak'],
             rf.get('frac'), rf['hbm']['frac'], rf.get('per_ply', {}).get('kernel'), rf.get('per_ply', {}).get('frac')))
md.append('## rocprofv3 --kernel-trace --stats (same command, extras and CPU baseline off)\n')
# one row per launch SHAPE (kernel x grid): rocprofv3's own stats file averages over every launch of a kernel NAME, and a bench
# run launches most kernels in several shapes (round 5's `k_children3` row averaged 8 192-parent launches with smaller ones and
# read as 8.2 TB/s)
all_trace = list(csv.DictReader(open(os.path.join(src, 'kt', 'kt_kernel_trace.csv'))))
shapes = collections.defaultdict(list)
for r in all_trace:
    shapes[(r['Kernel_Name'], int(r['Grid_Size_X']), int(r.get('Workgroup_Size_X') or 0))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
total_ns = float(sum(sum(v) for v in shapes.values())) or 1.0
md.append('| kernel | grid (threads) x workgroup | calls | avg ns | total % |\n|---|---|---|---|---|')
for (name, grid, wg), d in sorted(shapes.items(), key=lambda kv: -sum(kv[1]))[:8]:
    md.append('| `%s` | %d x %d | %d | %.0f | %.1f |' % (name[:80], grid, wg, len(d), mean(d), 100.0 * sum(d) / total_ns))
trace = [r for r in all_trace if kshort in r['Kernel_Name']]
full = max(int(r['Grid_Size_X']) for r in trace)
fullrows = [r for r in trace if int(r['Grid_Size_X']) == full]
durs = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in fullrows]
md.append('\nbench.py live launch_ms (HIP events over the 20 timed launches) = %.4f; rocprofv3 average of the %d full-batch '
          'launches of `%s` (burn-in, warm-up, timed: %d plies each) = %.4f ms (min %.4f, max %.4f).'
          % (rf['launch_ms'], len(durs), kernel, F, mean(durs) / 1e6, min(durs) / 1e6, max(durs) / 1e6))

# ---- calibration of the traffic counters on launches with known byte counts (tools/calib_traffic.py)
cal = {}
calB = 65536
known = {'k_unpack': {'read': 232 * calB, 'write': 2166 * calB}, 'k_pack': {'read': (3 * 361 + 3) * calB, 'write': 232 * calB}}
try:
    for cname, sub in (('FETCH_SIZE', 'cal_fetch'), ('WRITE_SIZE', 'cal_write')):
        for k in known:
            vals = [d[cname] for d in counters(sub, k, skip=1)]
            cal.setdefault(k, {})[cname] = mean(vals) * 1024
    md.append('\n## counter calibration (tools/calib_traffic.py: 65 536 boards, known bytes per launch)\n')
    md.append('| kernel | known read | FETCH_SIZE x 1 KiB | ratio | known write | WRITE_SIZE x 1 KiB | ratio |\n|---|---|---|---|---|---|---|')
    for k in known:
        md.append('| `%s` | %.1f MB | %.1f MB | %.3f | %.1f MB | %.1f MB | %.3f |' % (
            k, known[k]['read'] / 1e6, cal[k]['FETCH_SIZE'] / 1e6, cal[k]['FETCH_SIZE'] / known[k]['read'],
            known[k]['write'] / 1e6, cal[k]['WRITE_SIZE'] / 1e6, cal[k]['WRITE_SIZE'] / known[k]['write']))
    fetch_scale = known['k_pack']['read'] / cal['k_pack']['FETCH_SIZE']      # wide coalesced reads of byte planes
    write_scale = known['k_unpack']['write'] / cal['k_unpack']['WRITE_SIZE']  # the rollout kernel's own store pattern
    md.append('\ncorrection factors used below: FETCH_SIZE x %.3f (the guide prescribes x2 for wide coalesced reads on gfx950), '
              'WRITE_SIZE x %.3f (calibrated on `k_unpack`, the same emitter as the write-back of `%s`)' % (fetch_scale, write_scale, kshort))
except Exception as e:   # calibration pass missing (a LIGHT pass): the factors of the round's calibrated record, else the guide's
    fetch_scale, write_scale = 2.0, 1.0
    try:
        prev = [r for r in json.load(open(os.path.join(dst, 'pmc_rollout.json')))['records'] if r.get('fetch_scale')]
        fetch_scale, write_scale = prev[-1]['fetch_scale'], prev[-1]['write_scale']
        md.append('\n(no calibration pass in this run: the correction factors of %s are used - FETCH_SIZE x %.3f, WRITE_SIZE x %.3f)'
                  % (prev[-1]['source'].split(' ')[0], fetch_scale, write_scale))
    except Exception:
        md.append('\n(calibration pass not available: %s; using FETCH_SIZE x2, WRITE_SIZE x1)' % e)

# ---- config 5 under the same counters (the calibration script also expands 8 192 parents)
children_rec = None
try:
    cf = mean(d['FETCH_SIZE'] for d in counters('cal_fetch', 'k_children3', skip=1)) * 1024 * fetch_scale
    cw = mean(d['WRITE_SIZE'] for d in counters('cal_write', 'k_children3', skip=1)) * 1024 * write_scale
    algo = 8192 * (6 * 361 + 362 * 6 * 361)
    md.append('\n## config 5 (`k_children3<19, false, false>`, 8 192 parents): HBM traffic per launch\n')
    md.append('FETCH_SIZE -> %.1f MB read, WRITE_SIZE -> %.1f MB written (same correction factors); the algorithm moves %.1f MB '
              '(1 444 B in, 362 x 2 166 B out per parent): traffic / algorithmic = %.3f - every byte is written once, nothing is re-read.'
              % (cf / 1e6, cw / 1e6, algo / 1e6, (cf + cw) / algo))
    children_rec = {'kernel': 'k_children3<19, false, false>', 'parents': 8192, 'fetch_bytes': round(cf), 'write_bytes': round(cw),
                    'hbm_bytes_per_launch': round(cf + cw), 'algorithmic_bytes_per_launch': algo,
                    'source': 'profiles/%s_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/calib_traffic.py)' % tag}
except Exception as e:
    md.append('\n(config 5 traffic not collected: %s)' % str(e)[:120])

steps = games * F
traffic = {}
for name, sub in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write')):
    traffic[name] = mean(d[name] for d in counters(sub, kshort))
fetch_b = traffic['FETCH_SIZE'] * 1024 * fetch_scale
write_b = traffic['WRITE_SIZE'] * 1024 * write_scale
fused = rf['hbm']['algorithmic_bytes_per_launch']
md.append('\n## HBM traffic per launch (PMC, separate passes)\n')
md.append('FETCH_SIZE = %.0f KiB -> %.1f MB read; WRITE_SIZE = %.0f KiB -> %.1f MB written; total %.1f MB per launch = %.0f B '
          'per game per launch.  A fused launch must move %.1f MB (board in + board out + generator per game): traffic / '
          'algorithmic = %.2f.  At %.3f ms per launch that is %.0f GB/s = %.4f of the 8 TB/s peak - the kernel is not HBM-bound.'
          % (traffic['FETCH_SIZE'], fetch_b / 1e6, traffic['WRITE_SIZE'], write_b / 1e6, (fetch_b + write_b) / 1e6,
             (fetch_b + write_b) / games, fused / 1e6, (fetch_b + write_b) / fused, rf['launch_ms'],
             (fetch_b + write_b) / (rf['launch_ms'] * 1e-3) / 1e9, (fetch_b + write_b) / (rf['launch_ms'] * 1e-3) / 8e12))

md.append('\n## instruction mix per env step (PMC / (games x plies))\n')
mix = {}
for sub in ('pmc_inst', 'pmc_act', 'pmc_busy'):
    try:
        c = counters(sub, kshort)
    except Exception as e:      # (pmc_busy: derived / newer counters that a stack may not offer)
        md.append('- (%s not collected: %s)' % (sub, str(e)[:80]))
        continue
    keys = [k for k in c[0] if not k.startswith('_')]
    avg = {k: mean(d[k] for d in c) / steps for k in keys}
    mix.update(avg)
    md.append('- ' + ', '.join('%s %.2f' % (k, v) for k, v in sorted(avg.items())))
    md.append('  (VGPRs %s, LDS %s B per 64-thread workgroup)' % (c[0]['_vgpr'], c[0]['_lds']))
md.append('\nSQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* are documented as quad-cycles.  bench.py prints achieved = VALU instructions '
          'per step x env steps/s against a peak of one wave64 VALU instruction per SIMD every 2 cycles (SALU / LDS beside it).')
# ---- how busy is the VALU pipe, from counters (SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE x SIMDs))
valu_busy = None
try:
    xcds, simds = 8, 256 * 4
    act, gui = mix['SQ_ACTIVE_INST_VALU'], mix['GRBM_GUI_ACTIVE']
    per_xcd = gui / xcds            # the collected value is the sum over the 8 XCDs (= 8 x launch cycles)
    quad = act * 4.0 / (per_xcd * simds)
    valu_busy = {'formula': 'SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs)',
                 'sq_active_inst_valu_per_step': round(act, 3), 'grbm_gui_active_per_step_sum_of_8_xcds': round(gui, 4),
                 'value': round(quad, 4)}
    note = ('SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU on this kernel (%.2f vs %.2f per step): the counter advances once per VALU '
            'instruction, so the formula prices every instruction at a full quad-cycle' % (act, mix['SQ_INSTS_VALU']))
    if quad > 1.0:
        note += ' and comes out above 1 - the simple ops of this kernel issue in 2 cycles (profiles/r01_ubench_valu_rates.txt)'
    valu_busy['note'] = note
    if 'VALUBusy' in mix:
        valu_busy['rocprof_VALUBusy_percent'] = round(mix['VALUBusy'] * steps, 2)    # a per-launch percentage, not per step
    md.append('\n## VALU busy from counters\n')
    md.append('%s = **%.3f**.  %s.' % (valu_busy['formula'], quad, note))
except Exception as e:
    md.append('\n(VALU-busy counters not available: %s)' % e)

# ---- the record bench.py reads
pmc_path = os.path.join(dst, 'pmc_rollout.json')
try:
    allrec = json.load(open(pmc_path)).get('records', [])
except Exception:
    allrec = []
rec = {'kernel': kernel, 'size': N, 'plies_per_launch': F, 'games': games,
       'instr_per_step': {'valu': round(mix['SQ_INSTS_VALU'], 3), 'salu': round(mix['SQ_INSTS_SALU'], 3),
                          'lds': round(mix['SQ_INSTS_LDS'], 3)},
       'hbm_bytes_per_launch': round(fetch_b + write_b), 'fetch_bytes': round(fetch_b), 'write_bytes': round(write_b),
       'fetch_scale': round(fetch_scale, 4), 'write_scale': round(write_scale, 4),
       'source': 'profiles/%s_summary.md (rocprofv3 --pmc passes of `bench.py --plies-per-step %d`, tools/profile_round.sh)' % (tag, F)}
# the machine code these counters were collected on (the in-tree library travelled to the GPU box as it is here): bench.py
# recomputes the hash from the library it runs and says `pmc_stale` when the kernel has been rebuilt differently since
sys.path.insert(0, ROOT)
import bench as _bench   # noqa: E402
rec['kernel_code_sha16'] = _bench.kernel_code_hash(_bench.rollout_symbol_prefix(kernel) or '\0')
if valu_busy:
    rec['valu_busy'] = valu_busy
# static issue-cycle estimate of the ply loop (tools/isa_mix.py), when this round recorded one
try:
    mixtxt = open(os.path.join(dst, '%s_isa_mix.txt' % tag)).read()
    import re
    rec['valu_issue_cycles_per_instr'] = float(re.search(r'\(([0-9.]+) cycles per VALU instruction', mixtxt).group(1))
    rec['valu_issue_cycles_source'] = 'profiles/%s_isa_mix.txt (tools/isa_mix.py: static mix of the ply loop x measured issue rates)' % tag
except Exception:
    pass
allrec = [r for r in allrec if not (r['kernel'] == kernel and r['size'] == N and r['plies_per_launch'] == F and r['games'] == games)]
allrec.append(rec)
blob = {'records': allrec}
try:
    blob['children'] = json.load(open(pmc_path)).get('children')
except Exception:
    pass
if children_rec:
    children_rec['kernel_code_sha16'] = _bench.kernel_code_hash(_bench.CHILDREN_SYMBOL_PREFIX)
    blob['children'] = children_rec
json.dump(blob, open(pmc_path, 'w'), indent=1)

ops_stats = os.path.join(src, 'kt_ops', 'kt_kernel_stats.csv')
if os.path.exists(ops_stats):
    shutil.copy(ops_stats, os.path.join(dst, tag + '_ops_kernel_stats.csv'))
    shutil.copy(os.path.join(src, 'ops.json'), os.path.join(dst, tag + '_ops.json'))
    md.append('\n## the other entry points (`tools/bench_ops.py`, rocprofv3 --kernel-trace --stats; %s_ops.json has the rates)\n' % tag)
    ops_trace = os.path.join(src, 'kt_ops', 'kt_kernel_trace.csv')
    if os.path.exists(ops_trace):     # one row per launch shape, as above
        oshapes = collections.defaultdict(list)
        for r in csv.DictReader(open(ops_trace)):
            oshapes[(r['Kernel_Name'], int(r['Grid_Size_X']), int(r.get('Workgroup_Size_X') or 0))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        md.append('| kernel | grid (threads) x workgroup | calls | avg ns |\n|---|---|---|---|')
        for (name, grid, wg), d in sorted(oshapes.items(), key=lambda kv: -sum(kv[1]))[:24]:
            md.append('| `%s` | %d x %d | %d | %.0f |' % (name[:80], grid, wg, len(d), mean(d)))
    else:
        md.append('| kernel | calls | avg ns (all launch shapes of the name together) |\n|---|---|---|')
        for r in list(csv.DictReader(open(ops_stats)))[:16]:
            md.append('| `%s` | %s | %.0f |' % (r['Name'][:80], r['Calls'], float(r['AverageNs'])))
open(os.path.join(dst, tag + '_summary.md'), 'w').write('\n'.join(md) + '\n')
print('\n'.join(md))
