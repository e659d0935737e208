"""per-pair kernels vs the cap on their grid (A/B build, GG_AB_GRID_CAP): 8 192 workgroups of 4 pairs vs finer ones"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    for cap in ('8192', '3072', '4096', '16384', '32768'):
        env = dict(os.environ, GG_AB_GRID_CAP=cap, LIB='libgymgo_ab.so')
        print('cap', cap, subprocess.run([sys.executable, __file__, 'run'], env=env, capture_output=True, text=True).stdout.strip(), flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', os.environ['LIB'])
from gymgo_amd import gogame, state_utils
N, B = 19, 65536
cap = os.environ.pop('GG_AB_GRID_CAP')
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
ch = B // 16
for g in range(1, 16):
    gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 40, True)
gogame.batch_rollout(st, rng, 256 * 7, True)
os.environ['GG_AB_GRID_CAP'] = cap
def ev(fn, reps=24):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
out = (torch.empty(B, dtype=torch.float32, device='cuda'), torch.empty(B, dtype=torch.uint8, device='cuda'),
       torch.empty(B, dtype=torch.int32, device='cuda'), torch.empty(B, dtype=torch.int32, device='cuda'))
w = st.clone()
print('env_step bytes %.1f us' % ev(lambda: gogame.batch_env_step(w, None, rng, 7.5, 'real', True, out=out)), end='  ')
print('rollout 1 ply %.1f us' % ev(lambda: gogame.batch_rollout(w, rng, 1, True)), end='  ')
print('invalid_mask %.1f us' % ev(lambda: state_utils.batch_compute_invalid_moves(st, None, None)), end='  ')
print('track %.1f us' % ev(lambda: gogame.batch_track(st)), end='  ')
pk = gogame.batch_pack(st); acts = gogame.batch_sample_actions(st, rng)
print('next_states_packed %.1f us' % ev(lambda: gogame.batch_next_states_packed(pk, acts, check=False)), end='  ')
print('untrack %.1f us' % ev(lambda: gogame.batch_untrack(gogame.batch_track(st))))
