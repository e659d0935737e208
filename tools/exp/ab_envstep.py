"""One-ply env step on tracked boards for ONE library (LIB=<path relative to the repo root>, default: the shipped one):
uniform draw, policy-weighted draw with float32 / bfloat16 / float16 weights, with and without the observation.
    LIB=ab_libs/libgymgo_c2.so python tools/exp/ab_envstep.py; python tools/exp/ab_envstep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['LIB'])
from gymgo_amd import gogame

N, B = int(os.environ.get('GGN', 19)), int(os.environ.get('GGB', 65536))


def timed(fn, reps=40):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 3)
gogame.batch_rollout(st, rng, 250, True)
tracked = gogame.batch_track(st)
obs = torch.empty_like(st)
out = (torch.empty(B, dtype=torch.float32, device='cuda'), torch.empty(B, dtype=torch.uint8, device='cuda'),
       torch.empty(B, dtype=torch.int32, device='cuda'), torch.empty(B, dtype=torch.int32, device='cuda'))
w32 = torch.rand((B, N * N + 1), dtype=torch.float32, device='cuda')
rows = []
for name, kw in (('uniform', {}), ('weighted f32', {'weights': w32}), ('weighted bf16', {'weights': w32.to(torch.bfloat16)}),
                 ('weighted f16', {'weights': w32.to(torch.float16)})):
    t_obs = timed(lambda: gogame.batch_env_step_tracked(tracked, None, rng, 7.5, 'real', True, out=out, states_out=obs, **kw))
    t_no = timed(lambda: gogame.batch_env_step_tracked(tracked, None, rng, 7.5, 'real', True, out=out, **kw))
    rows.append((name, t_obs, t_no))
base = rows[0][1]
print('%-24s %s' % (os.environ.get('LIB', 'shipped'), '  '.join('%s %.1f us (%.2fx uniform; no obs %.1f)' % (n, a, base / a, b) for n, a, b in rows)), flush=True)
acts = torch.empty(B, dtype=torch.int32, device='cuda')
t = timed(lambda: gogame.batch_sample_weighted(st, w32, rng))
print('%-24s gg_batch_sample_weighted f32 %.1f us' % (os.environ.get('LIB', 'shipped'), t), flush=True)
