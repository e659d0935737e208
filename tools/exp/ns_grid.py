"""gg_batch_next_states vs its grid size (A/B build, GG_AB_NS_GRID): whole iterations per wave vs resident waves"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    for g in ('3072', '2979', '3277', '3328', '3641', '4096', '2731', '2048'):
        env = dict(os.environ, GG_AB_NS_GRID=g, LIB='libgymgo_ab.so')
        print('grid', g, subprocess.run([sys.executable, __file__, 'run'], env=env, capture_output=True, text=True).stdout.strip(), flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', os.environ['LIB'])
from gymgo_amd import gogame
N, B = 19, 65536
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
ch = B // 16
for g in range(1, 16):
    gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 40, True)
gogame.batch_rollout(st, rng, 256 * 7, True)
acts = gogame.batch_sample_actions(st, rng)
nxt, status = torch.empty_like(st), torch.empty(B, dtype=torch.int32, device='cuda')
fn = lambda: gogame.batch_next_states(st, acts, check=False, out=nxt, status=status)
fn(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(32): fn()
b.record(); torch.cuda.synchronize()
print('%.1f us' % (a.elapsed_time(b) / 32 * 1e3))
