"""shader-clock share of the phases of k_next_states32 (a -DGG_AB -DGG_AB_PROF build: tools/exp/libgymgo_PROF.so)"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', 'libgymgo_PROF.so')
from gymgo_amd import gogame
L = ctypes.CDLL(_lib.LIB_PATH)
L.gg_ab_prof_read.argtypes = [ctypes.c_void_p]; L.gg_ab_prof_read.restype = ctypes.c_int32
N, B = 19, int(os.environ.get('B', '65536'))
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
ch = B // 16
for g in range(1, 16):
    gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 40, True)
gogame.batch_rollout(st, rng, 256 * 7, True)
acts = gogame.batch_sample_actions(st, rng)
nxt, status = torch.empty_like(st), torch.empty(B, dtype=torch.int32, device='cuda')
gogame.batch_next_states(st, acts, check=False, out=nxt, status=status)
buf = (ctypes.c_ulonglong * 8)()
L.gg_ab_prof_read(buf)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(8): gogame.batch_next_states(st, acts, check=False, out=nxt, status=status)
b.record(); torch.cuda.synchronize()
L.gg_ab_prof_read(buf)
v = list(buf); tot = sum(v)
names = ['load + placement', 'seeds (x11)', 'floods (x11)', 'counters (x11)', 'captures + patch', 'mask', 'emit', '-']
waves = (B + 31) // 32
print('B %d: %.1f us per launch (instrumented); clock ticks per wave-iteration: %.0f' % (B, a.elapsed_time(b) / 8 * 1e3, tot / 8 / waves))
for n, x in zip(names, v):
    print('  %-20s %5.1f %%   %8.0f ticks per wave' % (n, 100.0 * x / max(tot, 1), x / 8 / waves))
