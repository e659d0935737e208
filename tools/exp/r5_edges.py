"""Where k_rollout5 takes over from k_rollout4 (A/B build, GG_AB_R5 = 0 / 1): games per launch between one and two waves of
32 boards per SIMD, and launch lengths below 8 plies.    LIB=ab_tmp/libgymgo_ab.so python tools/exp/r5_edges.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], 'none']
import importlib.util
spec = importlib.util.spec_from_file_location('r5c', os.path.join(ROOT, 'tools', 'exp', 'r5_check.py'))
m = importlib.util.module_from_spec(spec)
try:
    spec.loader.exec_module(m)
except SystemExit:
    pass
for B, F in ((34816, 256), (36864, 256), (40960, 256), (45056, 256), (65536, 2), (65536, 3), (65536, 4), (65536, 6), (131072, 4)):
    row = []
    for r5 in (False, True):
        ms, dg = m.rate(B, r5, F, reps=12 if F >= 64 else 80)
        row.append('%s %.4f ms %s' % ('r5' if r5 else 'r4', ms, dg))
    print('B %6d x %3d plies: ' % (B, F) + ' | '.join(row), flush=True)
