"""finish times of the workgroups of one gg_batch_next_states launch (-DGG_AB_WHERE build)"""
import os, sys, ctypes, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', 'libgymgo_where.so')
from gymgo_amd import gogame
L = ctypes.CDLL(_lib.LIB_PATH)
N, B = 19, 65536
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
ch = B // 16
for g in range(1, 16):
    gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 40, True)
gogame.batch_rollout(st, rng, 256 * 7, True)
acts = gogame.batch_sample_actions(st, rng)
nxt, status = torch.empty_like(st), torch.empty(B, dtype=torch.int32, device='cuda')
for _ in range(3): gogame.batch_next_states(st, acts, check=False, out=nxt, status=status)
n = 3072
buf = (ctypes.c_uint * (3 * n))()
L.gg_ab_where_read(buf, n)
a = np.frombuffer(buf, dtype=np.uint32).reshape(n, 3)
xcc, hw, dur = a[:, 0] & 0xF, a[:, 1], a[:, 2].astype(np.float64) / 100.0
key = ((xcc.astype(np.int64) << 16) | (((hw >> 13) & 7).astype(np.int64) << 8) | ((hw >> 8) & 15)) << 2 | ((hw >> 4) & 3)
per = collections.defaultdict(list)
for k, d in zip(key, dur): per[k].append(d)
rows = np.array([sorted(v)[:3] for v in per.values() if len(v) >= 3])
print('k_next_states2, %d workgroups: duration min %.1f mean %.1f max %.1f us; per SIMD sorted (mean over %d SIMDs): %s' % (n, dur.min(), dur.mean(), dur.max(), len(rows), np.round(rows.mean(axis=0), 1).tolist()))
ids = collections.defaultdict(list)
for i, (k, d) in enumerate(zip(key, dur)): ids[k].append((d, i))
shown = 0
ok = 0
for k, v in ids.items():
    v.sort()
    order = [i for _, i in v]
    if order == sorted(order): ok += 1
    if shown < 6:
        print('  SIMD %x: (blockIdx, us) by finish time: %s' % (k, [(i, round(d, 1)) for d, i in v])); shown += 1
print('  SIMDs whose finish order == blockIdx order: %d of %d; blockIdx // 1024 distinct per SIMD: %d' % (ok, len(ids), sum(len(set(i // 1024 for _, i in v)) == len(v) for v in ids.values())))
