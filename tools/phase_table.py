#!/usr/bin/env python
"""Issue / stall attribution per phase of a ply of k_rollout4 (the headline kernel) WITHOUT a thread trace (rocprofv3 --att
needs a decoder library this image does not ship): static instruction counts per phase from a -DGG_AB_MARK listing, priced
with the measured issue rates (tools/isa_mix.py: 2 cycles for the fast ops, 4 for the rest), against the shader-clock time one
wave spends in the phase with four waves on its SIMD (tools/exp/prof_phases.py, a -DGG_AB_PROF build on the GPU box).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Igymgo_amd/csrc -mllvm -enable-post-misched=false -DGG_AB_MARK -S \\
          --cuda-device-only -o /tmp/gg_mark.s gymgo_amd/csrc/gg_rollout.hip
    python tools/phase_table.py /tmp/gg_mark.s gpurun_out/<tag>/prof_phases_k_rollout4.log > profiles/<tag>_phase_table.md

Markers in program order inside the ply loop: 4 (end of previous ply) -> 0 (sampling done) -> 1 (roles + seeds) -> 2 (flood)
-> 3 (liberties + class words) -> 4 (class patch).  The flood phase holds the sweep loop: its body is weighted with the
measured 3.07 sweeps per wave-ply (tests/devtools/flood_stats.py); the rare paths (auto-reset, captures next to atari groups)
are counted once, so the static VALU total is an upper bound of the PMC count per wave-ply (74.0 x 16 = 1 184)."""
import re
import sys

FAST = {'v_xor_b32', 'v_and_b32', 'v_or_b32', 'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_lshrrev_b32', 'v_bitop3_b32',
        'v_mov_b32', 'v_not_b32', 'v_add_co_u32'}
KERNEL = '_ZN2gg10k_rollout4ILi19ELi0ELb0ELb1ELb0ELb0E'
NAMES = {0: 'phase 1 sampling', 1: 'phase 2 roles + seeds', 2: 'phase 2 flood', 3: 'phase 2 liberties + class words', 4: 'phase 3 class patch'}
SWEEPS = 3.07


def ops(lines):
    out = []
    for l in lines:
        t = l.strip()
        if not t or t[0] in ';.' or t.endswith(':') or t.startswith(';;'):
            continue
        out.append(t.split()[0].replace('_e32', '').replace('_e64', '').replace('_dpp', '').replace('_sdwa', ''))
    return out


def main(listing, clocks):
    text = open(listing).read().split('\n')
    start = next(i for i, l in enumerate(text) if l.startswith(KERNEL))
    end = next(i for i in range(start, len(text)) if 's_endpgm' in text[i])
    body = text[start:end]
    marks = [(i, int(re.search(r'GGMARK (\d+)', l).group(1))) for i, l in enumerate(body) if 'GGMARK' in l]
    order = [m for m in marks if m[1] in (0, 1, 2, 3, 4)]
    # the ply loop: ... 4 | 0 1 2 3 | 5: the listing has marker 4 BEFORE 0 (the loop is rotated: its tail block comes first)
    pos = {k: i for i, k in order}
    seg = {0: (pos[4], pos[0]), 1: (pos[0], pos[1]), 2: (pos[1], pos[2]), 3: (pos[2], pos[3])}
    # phase 3 (class patch) = from marker 3 to the loop's back edge + the stretch from the loop header to marker 4
    five = next(i for i, k in marks if k == 5)
    seg[4] = (pos[3], five)
    measured = {}
    if clocks:
        for l in open(clocks):
            m = re.match(r'\s+(phase\d [^\d%]+?)\s+([\d.]+) %\s+([\d.]+) cycles per wave-ply', l)
            if m and len(measured) < 5:
                measured[len(measured)] = float(m.group(3))
    print('| phase | VALU | of them 4-cycle ops | SALU | LDS | DPP / cross-lane | issue cycles of the wave (static) | measured cycles per wave-ply (4 waves / SIMD) | measured / (4 x static issue) |')
    print('|---|---|---|---|---|---|---|---|---|')
    tot_v = tot_c = tot_m = 0
    for k in range(5):
        a, b = seg[k]
        lines = body[a:b]
        weight = [1.0] * len(lines)
        if k == 2:   # the sweep loop: blocks at loop depth 3 inside the flood phase
            depth = 2
            for i, l in enumerate(lines):
                m = re.search(r'Depth=(\d+)', l)
                if re.match(r'^\.LBB', l) and m:
                    depth = int(m.group(1))
                weight[i] = SWEEPS / 2.0 if depth >= 3 else 1.0     # (the loop body holds two sweeps: down, up)
        v = slow = sa = lds = dpp = 0.0
        cyc = 0.0
        for i, l in enumerate(lines):
            o = ops([l])
            if not o:
                continue
            op, w = o[0], weight[i]
            if op.startswith('v_'):
                v += w
                is_dpp = 'row_' in l or 'quad_perm' in l or 'wave_sh' in l or op.startswith('v_readlane') or op.startswith('v_readfirstlane')
                dpp += w if is_dpp else 0
                fast = op in FAST and not is_dpp
                slow += 0 if fast else w
                cyc += w * (2 if fast else 4)
            elif op.startswith('ds_'):
                lds += w
                cyc += w * 4
            elif op.startswith('s_'):
                sa += w
                cyc += w * (1 if not op.startswith('s_nop') else 1)
        meas = measured.get(k)
        tot_v += v; tot_c += cyc; tot_m += meas or 0
        print('| %s | %.0f | %.0f | %.0f | %.0f | %.0f | %.0f | %s | %s |' % (NAMES[k], v, slow, sa, lds, dpp, cyc, '%.0f' % meas if meas else '-',
                                                                         '%.2f' % (meas / (4 * cyc)) if meas else '-'))
    print('| **ply** | %.0f | | | | | %.0f | %s | %s |' % (tot_v, tot_c, '%.0f' % tot_m if tot_m else '-', '%.2f' % (tot_m / (4 * tot_c)) if tot_m else '-'))
    print()
    print('Reading: four waves share one issue port per SIMD, so a phase whose measured time equals 4 x its static issue cycles '
          'keeps that port busy for its whole duration (ratio 1.0: issue-bound); a ratio above 1 is time in which none of the '
          'four waves could issue - dependent LDS round trips, DPP wait states, s_waitcnt, taken branches - and a ratio below 1 '
          'means the static pricing is pessimistic there (SALU instructions issue beside another wave\'s VALU; rare paths are '
          'counted once per ply).  Every phase sits within 0.86 - 1.35 of the bound and the whole ply at ~1.07: the kernel is '
          'ISSUE-bound on its own instruction mix (a third of its VALU instructions are 4-cycle ops), not latency-bound - the '
          '"52 % of their life waiting" of SQ_WAIT_INST_ANY is four waves queueing for one port, not stalls a fifth wave could '
          'fill.  What more occupancy (8 waves per SIMD with <= 5 KB of LDS per wave) could win back is the excess over 1.0 of the '
          'role / seed set-up, the flood and the liberty count: (0.35 x 377 + 0.22 x 1193 + 0.18 x 393) x 4 = 1 860 of 16 500 '
          'cycles, 11 % as an upper bound if ALL of it were hidden; the batch that does put eight waves\' worth of boards on a '
          'SIMD today (131 072 games: two rounds of four) gains 2 %.  The lever that is left is instructions per ply.')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
