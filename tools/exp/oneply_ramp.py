"""When do the workgroups of a ONE-PLY byte-plane launch enter and leave the machine?  (-DGG_AB_PROF build.)  Per workgroup:
entry and exit on the 100 MHz wall clock -> the spread of the entries (the dispatcher's ramp), the time a workgroup is
resident, the span of the launch.  GGN / GGB; KERNEL=k_rollout2|k_rollout_lat."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, os.environ.get('LIB', 'ab_libs/libgg_prof.so'))
from gymgo_amd import gogame
L = ctypes.CDLL(_lib.LIB_PATH)
N, B, F = int(os.environ.get('GGN', 9)), int(os.environ.get('GGB', 4096)), int(os.environ.get('PLIES', 1))
kernel = os.environ.get('KERNEL', 'k_rollout2')
os.environ['GG_AB_LAT_MAX'] = '0' if kernel == 'k_rollout2' else str(1 << 30)
os.environ['GG_AB_LAT_PLIES'] = '1'
sfx = 'kernels' if kernel == 'k_rollout2' else 'lat'
rd, raw = getattr(L, 'gg_ab_prof_read_' + sfx), getattr(L, 'gg_ab_prof_raw_' + sfx)
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
ch = max(1, B // 16)
for g in range(1, 16):
    gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 8, True)
for _ in range(3): gogame.batch_rollout(st, rng, F, True)
buf10 = (ctypes.c_ulonglong * 10)()
SL = 16384
rawbuf = (ctypes.c_ulonglong * (10 * SL))()
GRAPH = int(os.environ.get('GRAPH', 0))   # > 0: the LAST of that many launches replayed as one hipGraph (no host in between)
graph = None
if GRAPH:
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side): gogame.batch_rollout(st, rng, F, True)
    side.synchronize()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(GRAPH): gogame.batch_rollout(st, rng, F, True)
    graph.replay(); torch.cuda.synchronize()
ramp, resid, span, lastq, deciles = [], [], [], [], []
for _ in range(16):
    rd(buf10)
    if graph is not None: graph.replay()
    else: gogame.batch_rollout(st, rng, F, True)
    raw(rawbuf, SL)
    a = np.frombuffer(rawbuf, dtype=np.uint64).reshape(SL, 10)
    a = a[a[:, 9] != 0]
    t0 = a[:, 8].min()
    ent = (a[:, 8] - t0).astype(np.float64) * 0.01
    ext = (a[:, 9] - t0).astype(np.float64) * 0.01
    ramp.append(np.percentile(ent, [50, 90, 99, 100]))
    deciles.append(np.percentile(ent, list(range(10, 100, 10))))
    resid.append(np.percentile(ext - ent, [10, 50, 90, 100]))
    span.append(ext.max())
    lastq.append(np.sort(ext)[-max(1, len(ext) // 100):].min())
ramp, resid = np.median(np.array(ramp), axis=0), np.median(np.array(resid), axis=0)
print('%s N %d B %d F %d (%d workgroups, %s): entries after the first one: median %.2f us, p90 %.2f, p99 %.2f, last %.2f | '
      'resident per workgroup: p10 %.2f us, median %.2f, p90 %.2f, max %.2f | 99 %% of the workgroups gone at %.2f us, span %.2f us'
      % (kernel, N, B, F, len(a), 'last of a %d-launch graph' % GRAPH if GRAPH else 'stand-alone launch', ramp[0], ramp[1], ramp[2], ramp[3],
         resid[0], resid[1], resid[2], resid[3], float(np.median(lastq)), float(np.median(span))), flush=True)
print('    entry deciles (us): ' + ' '.join('%.2f' % x for x in np.median(np.array(deciles), axis=0)), flush=True)
