"""Sweeps per flood batch of k_rollout5 on the stationary mix (A/B build with -DGG_AB_SWEEPS: make ab EXTRA=-DGG_AB_SWEEPS).
    LIB=ab_tmp/libgg_sweeps.so python tools/exp/r5_sweeps.py"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, os.environ.get('LIB', 'ab_tmp/libgg_sweeps.so'))
from gymgo_amd import gogame
L = ctypes.CDLL(_lib.LIB_PATH)
L.gg_ab_sweeps_read_r5.argtypes = [ctypes.c_void_p]; L.gg_ab_sweeps_read_r5.restype = ctypes.c_int32
N, F, B = 19, 256, 65536
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
ch = B // 16
for g in range(1, 16):
    gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 40, True)
gogame.batch_rollout(st, rng, 4 * F, True)
buf = (ctypes.c_ulonglong * 2)()
L.gg_ab_sweeps_read_r5(buf)
for _ in range(4): gogame.batch_rollout(st, rng, F, True)
L.gg_ab_sweeps_read_r5(buf)
print('k_rollout5: %.3f sweeps per flood batch (%d batches; %.4f batches per wave-ply)' % (buf[0] / buf[1], buf[1], buf[1] / (B / 32 * F * 4)))
