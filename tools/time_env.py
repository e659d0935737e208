import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gymgo_amd import gogame
B, N = 65536, 19
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 3)
for g in range(16):
    gogame.batch_rollout(st[g * 4096:(g + 1) * 4096], rng[g * 4096:(g + 1) * 4096], 150 + 20 * g, True)
keep = st.clone()
out = None
def fn():
    global out
    out = gogame.batch_env_step(st, None, rng, 7.5, 'real', True, out=out)
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40): fn()
e1.record(); torch.cuda.synchronize()
print('GG_DBG', os.environ.get('GG_DBG', '0'), 'SYNC', os.environ.get('GG_SYNC_IO', '0'), 'env_step %.1f us' % (e0.elapsed_time(e1) / 40 * 1e3))
