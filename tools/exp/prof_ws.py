import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', 'libgymgo_PROF.so')
from gymgo_amd import gogame
L = ctypes.CDLL(_lib.LIB_PATH)
L.gg_ab_prof_read.argtypes = [ctypes.c_void_p]; L.gg_ab_prof_read.restype = ctypes.c_int32
B, N = 65536, 19
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 3)
for g in range(16):
    gogame.batch_rollout(st[g*4096:(g+1)*4096], rng[g*4096:(g+1)*4096], 100 + 40*g, True)
status = torch.empty(B, dtype=torch.int32, device='cuda'); ws = gogame.next_states_workspace(B, N)
cur, nxt = st, torch.empty_like(st)
buf = (ctypes.c_ulonglong * 8)()
names = ['phase1', 'phase2 setup', 'flood', 'liberties', 'phase3', '-', 'load', 'write-back']
def report(tag):
    L.gg_ab_prof_read(buf); v = list(buf); tot = sum(v) or 1
    print(tag, ' '.join('%s %.0f%%' % (n, 100.0 * x / tot) for n, x in zip(names, v) if x), ' total clocks/wave %.0f' % (tot / 4096 / 16))
for it in range(3):
    a = gogame.batch_sample_actions(cur, rng)
    gogame.batch_next_states(cur, a, check=False, out=nxt, status=status, workspace=ws); cur, nxt = nxt, cur
L.gg_ab_prof_read(buf)
for it in range(16):
    a = gogame.batch_sample_actions(cur, rng)
    gogame.batch_next_states(cur, a, check=False, out=nxt, status=status, workspace=ws); cur, nxt = nxt, cur
report('next_states_ws (hits):')
tr = gogame.batch_track(cur); obs = torch.empty_like(cur)
out = (torch.empty(B, dtype=torch.float32, device='cuda'), torch.empty(B, dtype=torch.uint8, device='cuda'), torch.empty(B, dtype=torch.int32, device='cuda'), torch.empty(B, dtype=torch.int32, device='cuda'))
L.gg_ab_prof_read(buf)
for it in range(16):
    gogame.batch_env_step_tracked(tr, None, rng, 7.5, 'real', True, out=out, states_out=obs)
report('env_step_tracked + obs   :')
