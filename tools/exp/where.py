"""How long does a 1 024-ply launch of the fused rollout take on each part of the chip?  (-DGG_AB_WHERE build: every
workgroup records its XCC, CU, SIMD and its own duration)"""
import os, sys, ctypes, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', os.environ.get('LIB', 'libgymgo_where.so'))
from gymgo_amd import gogame
L = ctypes.CDLL(_lib.LIB_PATH)
N, F, B = 19, int(os.environ.get("F", "1024")), int(os.environ.get('B', '65536'))
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
ch = B // 16
for g in range(1, 16):
    gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 40, True)
gogame.batch_rollout(st, rng, 512, True)
gogame.batch_rollout(st, rng, F, True)
n = min(16384, B // 16)
buf = (ctypes.c_uint * (3 * n))()
L.gg_ab_where_read(buf, n)
a = np.frombuffer(buf, dtype=np.uint32).reshape(n, 3)
xcc, hw, dur = a[:, 0] & 0xF, a[:, 1], a[:, 2].astype(np.float64) / 100.0   # us
cu = (hw >> 8) & 15; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
print('B %d, %d workgroups, %d plies: duration per workgroup  min %.0f  mean %.0f  max %.0f us' % (B, n, F, dur.min(), dur.mean(), dur.max()))
for x in sorted(set(xcc)):
    d = dur[xcc == x]
    print('  XCC %d: %4d workgroups  mean %.0f us  min %.0f  max %.0f' % (x, len(d), d.mean(), d.min(), d.max()))
key = (xcc.astype(np.int64) << 16) | (se.astype(np.int64) << 8) | cu
percu = collections.defaultdict(list)
for k, d in zip(key, dur): percu[k].append(d)
m = np.array([np.mean(v) for v in percu.values()])
print('  per CU mean duration: min %.0f  median %.0f  max %.0f us over %d CUs' % (m.min(), np.median(m), m.max(), len(m)))
skey = (key << 2) | simd
per = collections.defaultdict(list)
for k, d in zip(skey, dur): per[k].append(d)
rows = np.array([sorted(v)[:4] + [np.nan] * (4 - len(v[:4])) for v in per.values() if len(v) >= 4 and B == 65536] or [[0, 0, 0, 0]])
print('  per SIMD, the four workgroups sorted by duration (mean over SIMDs): %s us' % np.round(np.nanmean(rows, axis=0)).tolist())
if os.environ.get('DUMP'):
    # several consecutive launches: is a workgroup's duration predictable from its group index / from the previous launch?
    durs = [dur.copy()]
    for _ in range(int(os.environ.get('LAUNCHES', '5'))):
        gogame.batch_rollout(st, rng, F, True)
        L.gg_ab_where_read(buf, n)
        durs.append(np.frombuffer(buf, dtype=np.uint32).reshape(n, 3)[:, 2].astype(np.float64) / 100.0)
    stones = (st[:, 0].sum(dim=(1, 2)) + st[:, 1].sum(dim=(1, 2))).reshape(n, 16).float().mean(dim=1).cpu().numpy()
    np.savez(os.path.join(ROOT, 'gpurun_out', os.environ['DUMP']), dur=np.stack(durs), hw=hw, xcc=xcc, stones=stones)
