"""persistent per-pair kernels: even shares on the old grid (GG_AB_EVEN) vs the resident grid split by wave age
(GG_AB_CUT3 / GG_AB_CUT4 = cumulative shares of a SIMD's pairs, oldest wave first); us per 65 536 boards, stationary mix"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CONFIGS = [
    {'GG_AB_EVEN': '1'},
    {},
]
SIZES = os.environ.get('GGN', '19').split(',')
if len(sys.argv) == 1:
    for n in SIZES:
        for cfg in CONFIGS:
            env = dict(os.environ, LIB='libgymgo_ab.so', GGN=n, **cfg)
            r = subprocess.run([sys.executable, __file__, 'run'], env=env, capture_output=True, text=True)
            print(n, cfg, '\n   ', r.stdout.strip() or r.stderr[-800:], flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', os.environ['LIB'])
from gymgo_amd import gogame, state_utils
N, B = int(os.environ.get('GGN', '19')), 65536
held = {k: os.environ.pop(k) for k in list(os.environ) if k.startswith('GG_AB_')}
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
ch = B // 16
for g in range(1, 16):
    gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * (N * N // 9), True)
gogame.batch_rollout(st, rng, 256 * 7, True)
os.environ.update(held)
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
print('env_step bytes %.1f' % ev(lambda: gogame.batch_env_step(w, None, rng, 7.5, 'real', True, out=out)), end='  ')
print('rollout 1 ply %.1f' % ev(lambda: gogame.batch_rollout(w, rng, 1, True)), end='  ')
print('invalid_mask %.1f' % ev(lambda: state_utils.batch_compute_invalid_moves(st, None, None)), end='  ')
print('track %.1f' % ev(lambda: gogame.batch_track(st)), end='  ')
pk = gogame.batch_pack(st); acts = gogame.batch_sample_actions(st, rng)
print('next_states_packed %.1f' % ev(lambda: gogame.batch_next_states_packed(pk, acts, check=False)), end='  ')
wp = pk.clone()
print('env_step packed %.1f' % ev(lambda: gogame.batch_env_step_packed(wp, None, rng, 7.5, 'real', True)), end='  ')
print('rollout packed 1 ply %.1f' % ev(lambda: gogame.batch_rollout_packed(wp, rng, 1, True)))
