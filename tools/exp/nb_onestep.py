import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', 'libgymgo_ab.so')
from gymgo_amd import gogame
B, N = 65536, 19
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 3)
for g in range(16):
    gogame.batch_rollout(st[g*4096:(g+1)*4096], rng[g*4096:(g+1)*4096], 100 + 40*g, True)
status = torch.empty(B, dtype=torch.int32, device='cuda'); ws = gogame.next_states_workspace(B, N)
passes = torch.full((B,), N * N, dtype=torch.int32, device='cuda')
cur, nxt = st.clone(), torch.empty_like(st)
def ev(fn, reps=30):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
def pp():
    global cur, nxt
    gogame.batch_next_states(cur, passes, check=False, out=nxt, status=status, workspace=ws); cur, nxt = nxt, cur
tr = gogame.batch_track(st); obs = torch.empty_like(st)
out = (torch.empty(B, dtype=torch.float32, device='cuda'), torch.empty(B, dtype=torch.uint8, device='cuda'), torch.empty(B, dtype=torch.int32, device='cuda'), torch.empty(B, dtype=torch.int32, device='cuda'))
print('NB=%s  next_states_ws (hit, passes) %.1f us   env sampled+obs %.1f us   env no obs %.1f us' % (os.environ.get('GG_AB_NB', 'auto'), ev(pp),
      ev(lambda: gogame.batch_env_step_tracked(tr, None, rng, 7.5, 'real', True, out=out, states_out=obs)),
      ev(lambda: gogame.batch_env_step_tracked(tr, None, rng, 7.5, 'real', True, out=out))))
