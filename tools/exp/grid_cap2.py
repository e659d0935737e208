"""streaming / sampling kernels vs the cap on their persistent grid (A/B build, GG_AB_GRID_CAP); us per 65 536 boards"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    for cap in ('8192', '2048', '4096', '16384', '32768', '65536'):
        env = dict(os.environ, GG_AB_GRID_CAP=cap, LIB='libgymgo_ab.so')
        r = subprocess.run([sys.executable, __file__, 'run'], env=env, capture_output=True, text=True)
        print('cap', cap, r.stdout.strip() or r.stderr[-600:], flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('LIB', 'shipped') != 'shipped':
    _lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', os.environ['LIB'])
from gymgo_amd import gogame
N, B = 19, 65536
cap = os.environ.pop('GG_AB_GRID_CAP', None)
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
ch = B // 16
for g in range(1, 16):
    gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 40, True)
gogame.batch_rollout(st, rng, 256 * 7, True)
if cap: os.environ['GG_AB_GRID_CAP'] = cap
def ev(fn, reps=24):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
w = torch.rand(B, N * N + 1, device='cuda')
tr = gogame.batch_track(st); pk = gogame.batch_pack(st)
ori = torch.randint(0, 8, (B,), dtype=torch.int32, device='cuda')
out_sym = torch.empty_like(st)
print('sample_weighted %.1f' % ev(lambda: gogame.batch_sample_weighted(st, w, rng)), end='  ')
print('sample_weighted_rows %.1f' % ev(lambda: gogame.batch_sample_weighted_rows(tr, N, w, rng)), end='  ')
print('symmetry %.1f' % ev(lambda: gogame.batch_symmetry(st, ori, out=out_sym)), end='  ')
print('symmetry_rows %.1f' % ev(lambda: gogame.batch_symmetry_rows(tr, N, ori)), end='  ')
print('pack %.1f' % ev(lambda: gogame.batch_pack(st)), end='  ')
print('unpack %.1f' % ev(lambda: gogame.batch_unpack(pk, N)), end='  ')
print('untrack %.1f' % ev(lambda: gogame.batch_untrack(tr, out=out_sym)), end='  ')
kids = torch.empty((8192, N * N + 1, 6, N, N), dtype=torch.uint8, device='cuda')
print('children8192 %.1f' % ev(lambda: gogame.batch_children(st[:8192], out=kids), reps=6))
