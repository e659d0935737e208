"""Static instruction mix of the multi-ply kernel's ply loop, priced with the measured issue rates of
profiles/r01_ubench_valu_rates*.txt (2 cycles per wave64 instruction for v_and/or/xor/add/sub/lshrrev/bitop3/mov, 4 for the
rest: v_bfrev, v_bcnt, v_cndmask, v_cmp, v_lshlrev, three-operand ops, DPP moves ...).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -mllvm -enable-post-misched=false -S --cuda-device-only -o /tmp/gg.s gymgo_amd/csrc/gg_rollout.hip
    python tools/isa_mix.py /tmp/gg.s > profiles/rNN_isa_mix.txt
    hipcc ... -DGG_AB_MARK -S ... -o /tmp/gg5.s gymgo_amd/csrc/gg_r5.hip      (default switches: the unit is built with them)
    python tools/isa_mix.py /tmp/gg5.s _ZN2gg10k_rollout5ILi19ELi0E 32      (k_rollout5: marker listing, boards per wave)

The ply loop is the longest stretch between two consecutive `v_mbcnt_lo` markers of k_rollout4<19, 0, false, true, false, false> (the kernel reads
its lane id afresh at the top of every ply and once more before the write-back).  Inner loops are weighted by the trip
counts measured on mid-game boards (tests/devtools/flood_stats.py: 3.07 flood sweeps per wave-ply on average; a capture on
some board of the wave on 85 % of the plies; the auto-reset block is rare)."""
import collections
import re
import sys

FAST = {'v_xor_b32', 'v_and_b32', 'v_or_b32', 'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_lshrrev_b32', 'v_bitop3_b32',
        'v_mov_b32', 'v_not_b32', 'v_add_co_u32'}
KERNEL = '_ZN2gg10k_rollout4ILi19ELi0ELb0ELb1ELb0ELb0E'


def block_depths(lines):
    """Loop depth of every line (the `Depth=` comments behind the block labels; 2 = the ply loop itself)."""
    depth, out = 2, []
    for i, l in enumerate(lines):
        if re.match(r'^\.LBB\d+_\d+:', l):
            found, j = re.findall(r'Depth=(\d+)', l), i + 1
            while j < len(lines) and lines[j].strip().startswith(';'):
                found += re.findall(r'Depth=(\d+)', lines[j])
                j += 1
            depth = int(found[-1]) if found else depth
        out.append(depth)
    return out


def price(lines, weights, KERNEL, NB, title):
    tot, cyc, slow = collections.Counter(), 0.0, collections.Counter()
    for l, w in zip(lines, weights):
        l = l.strip()
        if not l or l[0] in '.;' or l.endswith(':'):
            continue
        op = l.split()[0]
        base = re.sub(r'_e(32|64)$', '', op)
        if op.startswith('v_'):
            fast = base in FAST and 'dpp' not in l
            tot['valu fast (2 cycles)' if fast else 'valu slow (4 cycles)'] += w
            cyc += w * (2 if fast else 4)
            if not fast:
                slow[base] += w
        elif op.startswith('s_'):
            tot['salu'] += w
        elif op.startswith('ds_'):
            tot['lds'] += w
        else:
            tot['other'] += w
    print(title)
    for k in sorted(tot):
        print('  %-22s %8.1f per wave-ply  (%.1f per env step at %d boards per wave)' % (k, tot[k], tot[k] / NB, NB))
    valu = tot['valu fast (2 cycles)'] + tot['valu slow (4 cycles)']
    print('  VALU issue cycles per wave-ply: %.0f  (%.2f cycles per VALU instruction on average)' % (cyc, cyc / valu))
    print('  slow ops: ' + ', '.join('%s %.0f' % (k, v) for k, v in slow.most_common(12)))
    print('  => VALU pipe busy (static estimate) = roofline.frac x %.2f / 2 (bench.py prices every VALU instruction at the 2-cycle peak)' % (cyc / valu))


def main_marked(body, KERNEL, NB):
    """k_rollout5 (gg_v5.h), from a -DGG_AB_MARK listing: the ply loop is everything between markers 4 (end of the previous
    ply; the loop is rotated, its tail comes first) and 5; the flood's sweep loop is the deepest loop between markers 1 and 2
    (weighted with the measured sweeps per wave-ply), the auto-reset clear loop (between 4 and 0) and the atari re-flood after a
    capture (between 3 and 5) are rare; the job-batch loop runs once (a second batch on < 1 % of the plies)."""
    marks = {int(re.search(r'GGMARK (\d+)', l).group(1)): i for i, l in enumerate(body) if 'GGMARK' in l}
    a, b = marks[4], marks[5]
    loop = body[a:b]
    depth = block_depths(body)[a:b]
    pos = {k: marks[k] - a for k in (0, 1, 2, 3)}
    base = depth[pos[0]]                           # the ply loop's own depth (marker 0 sits in its straight-line body)
    weights = [1.0] * len(loop)
    floodd = max(depth[pos[1]:pos[2]])
    for i in range(len(loop)):
        if i < pos[0] and depth[i] > base:
            weights[i] = 0.02                      # auto-reset clear loop
        elif pos[1] <= i < pos[2] and depth[i] == floodd and floodd > base + 1:
            weights[i] = 3.07 / 2.0                # the sweep loop body holds two sweeps (down, up)
        elif i >= pos[3] and depth[i] > base:
            weights[i] = 0.3                       # atari re-flood
    price(loop, weights, KERNEL, NB, 'ply loop of %s... (marker listing): %d static lines, flood loop depth %d' % (KERNEL, len(loop), floodd))


def main(path, KERNEL=KERNEL, NB=16):
    text = open(path).read().split('\n')
    start = next(i for i, l in enumerate(text) if l.startswith(KERNEL))
    end = next(i for i in range(start, len(text)) if 's_endpgm' in text[i])
    body = text[start:end]
    if any('GGMARK' in l for l in body):
        return main_marked(body, KERNEL, NB)
    marks = [i for i, l in enumerate(body) if 'v_mbcnt_lo_u32_b32' in l]
    assert len(marks) >= 2, marks
    # the kernel reads its lane id afresh in several places (group loop, FairShare, ply, write-back): the ply body is the
    # longest stretch between two consecutive reads
    gaps = [(marks[i + 1] - marks[i], i) for i in range(len(marks) - 1)]
    i = max(gaps)[1]
    marks = [marks[i], marks[i + 1]]
    loop = body[marks[0]:marks[1]]
    # blocks of the loop in program order: (first line, label, depth)
    depth, blocks = 2, []
    for i, l in enumerate(loop):
        if re.match(r'^\.LBB\d+_\d+:', l):
            # "in Loop: Header=BBx_y Depth=d", or for a loop header several comment lines ending in "This ... Header: Depth=d"
            found, j = re.findall(r'Depth=(\d+)', l), i + 1
            while j < len(loop) and loop[j].strip().startswith(';'):
                found += re.findall(r'Depth=(\d+)', loop[j])
                j += 1
            depth = int(found[-1]) if found else 2
        blocks.append(depth)
    # depth-3 runs in program order: [0] auto-reset clear loop, [1] the flood, [2] the atari re-flood after a capture
    runs, prev = [], 2
    for i, d in enumerate(blocks):
        if d >= 3 and prev < 3:
            runs.append([i, i])
        if d >= 3:
            runs[-1][1] = i
        prev = d
    weights = [1.0] * len(loop)
    names = ['auto-reset clear (rare)', 'flood', 'atari re-flood after a capture']
    trip = [0.02, None, 0.3]
    for k, (a, b) in enumerate(runs[:3]):
        for i in range(a, b + 1):
            weights[i] = trip[k] if trip[k] is not None else 3.07 / 2.0   # the flood loop body holds two sweeps (down, up)
    tot, cyc = collections.Counter(), 0.0
    slow = collections.Counter()
    for l, w in zip(loop, weights):
        l = l.strip()
        if not l or l[0] in '.;' or l.endswith(':'):
            continue
        op = l.split()[0]
        base = re.sub(r'_e(32|64)$', '', op)
        if op.startswith('v_'):
            fast = base in FAST and 'dpp' not in l
            tot['valu fast (2 cycles)' if fast else 'valu slow (4 cycles)'] += w
            cyc += w * (2 if fast else 4)
            if not fast:
                slow[base] += w
        elif op.startswith('s_'):
            tot['salu'] += w
        elif op.startswith('ds_'):
            tot['lds'] += w
        else:
            tot['other'] += w
    print('ply loop of %s...: %d static lines; depth-3 runs: %s' % (KERNEL, len(loop), [(names[k], b - a + 1) for k, (a, b) in enumerate(runs[:3])]))
    for k in sorted(tot):
        print('  %-22s %8.1f per wave-ply  (%.1f per env step at %d boards per wave)' % (k, tot[k], tot[k] / NB, NB))
    valu = tot['valu fast (2 cycles)'] + tot['valu slow (4 cycles)']
    print('  VALU issue cycles per wave-ply: %.0f  (%.2f cycles per VALU instruction on average)' % (cyc, cyc / valu))
    print('  slow ops: ' + ', '.join('%s %.0f' % (k, v) for k, v in slow.most_common(12)))
    print('  => VALU pipe busy (static estimate) = roofline.frac x %.2f / 2 (bench.py prices every VALU instruction at the 2-cycle peak)' % (cyc / valu))


if __name__ == '__main__':
    main(sys.argv[1], *([sys.argv[2]] if len(sys.argv) > 2 else []), **({'NB': int(sys.argv[3])} if len(sys.argv) > 3 else {}))
