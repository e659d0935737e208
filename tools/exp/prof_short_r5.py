"""Phase clocks of k_rollout5 on SHORT tracked launches (A/B build with -DGG_AB_PROF, GG_AB_R5=1 forces the kernel): where the
fixed cost of a launch goes.   LIB=ab_tmp/libgg_prof.so python tools/exp/prof_short_r5.py"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, os.environ.get('LIB', 'ab_tmp/libgg_prof.so'))
from gymgo_amd import gogame
L = ctypes.CDLL(_lib.LIB_PATH)
rd = L.gg_ab_prof_read_r5; rd.argtypes = [ctypes.c_void_p]; rd.restype = ctypes.c_int32
N, B = 19, 65536
os.environ['GG_AB_R5'] = '0'
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
ch = B // 16
for g in range(1, 16):
    gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 40, True)
gogame.batch_rollout(st, rng, 1024, True)
tr = gogame.batch_track(st)
os.environ['GG_AB_R5'] = '1'
names = ['phase1 sampling', 'phase2 roles+setup', 'phase2 flood', 'phase2 liberties+cls', 'phase3 class patch', 'after the plies', 'load', 'write-back']
buf = (ctypes.c_ulonglong * 10)()
for F in (1, 4):
    for _ in range(10): gogame.batch_rollout_tracked(tr, rng, F, True)
    rd(buf)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): gogame.batch_rollout_tracked(tr, rng, F, True)
    b.record(); torch.cuda.synchronize()
    rd(buf)
    v = list(buf)[:8]
    waves = (B // 32) * 20
    print('tracked 65536 x %d plies: %.2f us per launch (instrumented)' % (F, a.elapsed_time(b) * 50))
    for n, x in zip(names, v):
        print('  %-22s %9.1f cycles per wave and launch' % (n, x / waves))
