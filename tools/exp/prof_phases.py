import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, os.environ.get('LIB', 'ab_libs/libgg_prof.so'))   # make -C gymgo_amd/csrc ab EXTRA=-DGG_AB_PROF
from gymgo_amd import gogame
L = ctypes.CDLL(_lib.LIB_PATH)
# (the fused launches with drawn moves live in gg_rollout.hip, whose phase clocks have their own reader)
L.gg_ab_prof_read = getattr(L, os.environ.get("PROF_READ", "gg_ab_prof_read_rollout"))
L.gg_ab_prof_read.argtypes = [ctypes.c_void_p]; L.gg_ab_prof_read.restype = ctypes.c_int32
N, F, B = 19, 256, int(os.environ.get('B', '65536'))
BPW = int(os.environ.get('BPW', '16'))   # boards per wave of the kernel that serves the launch (32: k_rollout5)
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
ch = B // 16
for g in range(1, 16):
    gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 40, True)
gogame.batch_rollout(st, rng, F, True)
buf = (ctypes.c_ulonglong * 10)()
L.gg_ab_prof_read(buf)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(4): gogame.batch_rollout(st, rng, F, True)
b.record(); torch.cuda.synchronize()
L.gg_ab_prof_read(buf)
v = list(buf)[:8]; tot = sum(v)
names = ['phase1 sampling', 'phase2 roles+setup', 'phase2 flood', 'phase2 liberties+cls', 'phase3 class patch', '-', 'load', 'write-back']
print('B %d: %.3f ms per launch (instrumented)' % (B, a.elapsed_time(b) / 4))
wave_plies = ((B + BPW - 1) // BPW) * 4 * F
for n, x in zip(names, v):
    print('  %-22s %5.1f %%  %8.1f cycles per wave-ply' % (n, 100.0 * x / tot, x / wave_plies))
