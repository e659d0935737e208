import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gymgo_amd import gogame, state_utils
B, N = 65536, 19
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 3)
for g in range(16):
    gogame.batch_rollout(st[g * 4096:(g + 1) * 4096], rng[g * 4096:(g + 1) * 4096], 150 + 20 * g, True)
def timed(fn, reps=100):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print('lib', os.environ.get('GYMGO_AMD_LIB', 'default').split('/')[-1], 'areas %.1f us' % timed(lambda: gogame.batch_areas(st)),
      'invalid_mask %.1f us' % timed(lambda: state_utils.batch_compute_invalid_moves(st, None, None)))
