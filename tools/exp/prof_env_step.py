"""Phase clocks of the tracked env step (k_rollout4<19, tracked, ENV>, gg_kernels.hip) and of a one-ply tracked rollout
(gg_rollout.hip) at 65 536 games (A/B build with -DGG_AB_PROF).   LIB=ab_tmp/libgg_prof.so python tools/exp/prof_env_step.py"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, os.environ.get('LIB', 'ab_tmp/libgg_prof.so'))
from gymgo_amd import gogame
L = ctypes.CDLL(_lib.LIB_PATH)
N, B = 19, 65536
os.environ['GG_AB_R5'] = '0'
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
ch = B // 16
for g in range(1, 16):
    gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 40, True)
gogame.batch_rollout(st, rng, 1024, True)
tr = gogame.batch_track(st)
obs = torch.empty_like(st)
names = ['phase1 sampling', 'phase2 roles+setup', 'phase2 flood', 'phase2 liberties+cls', 'phase3 class patch', 'after the plies', 'load', 'write-back + outputs']
buf = (ctypes.c_ulonglong * 10)()
out = None
def env_obs():
    global out
    out = gogame.batch_env_step_tracked(tr, None, rng, 7.5, 'real', True, out=out, states_out=obs)
out2 = None
def env_noobs():
    global out2
    out2 = gogame.batch_env_step_tracked(tr, None, rng, 7.5, 'real', True, out=out2)
w32 = torch.rand((B, N * N + 1), dtype=torch.float32, device='cuda'); w16 = w32.to(torch.bfloat16)
outw = None
def env_w(w):
    global outw
    outw = gogame.batch_env_step_tracked(tr, None, rng, 7.5, 'real', True, out=outw, states_out=obs, weights=w)
for label, fn, reader in (('env step + observation, float32 weights', lambda: env_w(w32), 'gg_ab_prof_read_kernels'), ('env step + observation, bfloat16 weights', lambda: env_w(w16), 'gg_ab_prof_read_kernels'),
                          ('env step + observation', env_obs, 'gg_ab_prof_read_kernels'), ('env step, no observation', env_noobs, 'gg_ab_prof_read_kernels'),
                          ('one-ply tracked rollout', lambda: gogame.batch_rollout_tracked(tr, rng, 1, True), 'gg_ab_prof_read_rollout')):
    rd = getattr(L, reader); rd.argtypes = [ctypes.c_void_p]; rd.restype = ctypes.c_int32
    for _ in range(10): fn()
    rd(buf)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    rd(buf)
    v = list(buf)[:8]
    waves = (B // 16) * 20
    print('%s: %.2f us per launch (instrumented)' % (label, a.elapsed_time(b) * 50))
    for n, x in zip(names, v):
        print('  %-22s %9.1f cycles per wave and launch' % (n, x / waves))
