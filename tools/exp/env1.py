import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('USE_AB'):
    _lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', 'libgymgo_ab.so')
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', os.environ['LIB'])
from gymgo_amd import gogame
B, N = 65536, 19
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 3)
for g in range(16):
    gogame.batch_rollout(st[g*4096:(g+1)*4096], rng[g*4096:(g+1)*4096], 100 + 40*g, True)
tr = gogame.batch_track(st); obs = torch.empty_like(st)
out = (torch.empty(B, dtype=torch.float32, device='cuda'), torch.empty(B, dtype=torch.uint8, device='cuda'),
       torch.empty(B, dtype=torch.int32, device='cuda'), torch.empty(B, dtype=torch.int32, device='cuda'))
def ev(fn, reps=40):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
acts = gogame.batch_sample_actions(st, rng).reshape(B, 1).contiguous()
pl = torch.empty(B, dtype=torch.int32, device='cuda')
tag = '%s NB=%s' % (os.environ.get('LIB', '-'), os.environ.get('GG_AB_NB', 'auto'))
print(tag, 'play_moves_tracked T=1   %.1f us' % ev(lambda: gogame.batch_play_moves_tracked(tr, acts, pl)))
print(tag, 'rollout_tracked F=1      %.1f us' % ev(lambda: gogame.batch_rollout_tracked(tr, rng, 1, True)))
print(tag, 'env sampled, no obs      %.1f us' % ev(lambda: gogame.batch_env_step_tracked(tr, None, rng, 7.5, 'real', True, out=out)))
print(tag, 'env sampled, obs         %.1f us' % ev(lambda: gogame.batch_env_step_tracked(tr, None, rng, 7.5, 'real', True, out=out, states_out=obs)))
a1 = acts.reshape(B)
print(tag, 'env given (stale), obs   %.1f us' % ev(lambda: gogame.batch_env_step_tracked(tr, a1, None, 7.5, 'real', True, out=out, states_out=obs)))
print(tag, 'env sampled heur, obs    %.1f us' % ev(lambda: gogame.batch_env_step_tracked(tr, None, rng, 7.5, 'heuristic', True, out=out, states_out=obs)))
print(tag, 'untrack                  %.1f us' % ev(lambda: gogame.batch_untrack(tr, out=obs)))
print(tag, 'rollout_tracked F=8      %.1f us' % ev(lambda: gogame.batch_rollout_tracked(tr, rng, 8, True)))
