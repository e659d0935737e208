"""Shader-clock share of the phases of a ply of k_rollout_lat as ONE wave experiences them (a -DGG_AB_PROF build:
LIB=ab_libs/libgg_prof.so; make -C gymgo_amd/csrc ab EXTRA=-DGG_AB_PROF).  GGN / GGB / PLIES as ab_small.py."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, os.environ.get('LIB', 'ab_libs/libgg_prof.so'))
from gymgo_amd import gogame
L = ctypes.CDLL(_lib.LIB_PATH)
L.gg_ab_prof_read = L.gg_ab_prof_read_lat     # (gg_lat.hip's own phase clocks)
L.gg_ab_prof_read.argtypes = [ctypes.c_void_p]; L.gg_ab_prof_read.restype = ctypes.c_int32
N, B, F = int(os.environ.get('GGN', 9)), int(os.environ.get('GGB', 4096)), int(os.environ.get('PLIES', 256))
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
ch = max(1, B // 16)
for g in range(1, 16):
    gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 8, True)
gogame.batch_rollout(st, rng, F, True)
buf = (ctypes.c_ulonglong * 10)()
L.gg_ab_prof_read(buf)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
reps = 4
for _ in range(reps): gogame.batch_rollout(st, rng, F, True)
b.record(); torch.cuda.synchronize()
L.gg_ab_prof_read(buf)
v = list(buf)[:8]; tot = sum(v)
per = (B + (4 if N <= 13 else 2) - 1) // (4 if N <= 13 else 2) * reps * F   # wave-plies
names = ['draw', 'tables (lut)', 'load bytes -> rows', '-', 'ply (lat_play)', '-', 'first classes / tracked load', 'write-back']
print('N %d B %d F %d: %.4f ms per launch (instrumented)' % (N, B, F, a.elapsed_time(b) / reps))
for n, x in zip(names, v):
    print('  %-26s %5.1f %%  %8.1f cycles per wave-ply' % (n, 100.0 * x / tot, x / per))
