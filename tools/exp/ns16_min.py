"""where should the sixteen-board kernels take over?  GG_AB_NS16_MIN = groups per SIMD from which they are used (A/B library)
vs the two-board kernels (GG_AB_NS16=0), per board size and batch size: next_states / env step / invalid mask, us per call"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    for cfg in ({'GG_AB_NS16': '0'}, {'GG_AB_NS16_MIN': '1'}):
        r = subprocess.run([sys.executable, __file__, 'run'], env=dict(os.environ, **cfg), capture_output=True, text=True)
        print(cfg, '\n' + (r.stdout.strip() or r.stderr[-500:]), flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', 'libgymgo_ab.so')
from gymgo_amd import gogame
held = {k: os.environ.pop(k) for k in list(os.environ) if k.startswith('GG_AB_')}
def ev(fn, reps=32):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for N in (19, 13, 9):
    line = []
    for B in (16384, 24576, 32768, 49152):
        st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
        ch = B // 16
        for g in range(1, 16):
            gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * (N * N // 9), True)
        gogame.batch_rollout(st, rng, 256 * 5, True)
        acts = gogame.batch_sample_actions(st, rng)
        out, status = torch.empty_like(st), torch.empty(B, dtype=torch.int32, device='cuda')
        eo = (torch.empty(B, dtype=torch.float32, device='cuda'), torch.empty(B, dtype=torch.uint8, device='cuda'),
              torch.empty(B, dtype=torch.int32, device='cuda'), torch.empty(B, dtype=torch.int32, device='cuda'))
        w = st.clone()
        os.environ.update(held)
        ns = ev(lambda: gogame.batch_next_states(st, acts, check=False, out=out, status=status))
        es = ev(lambda: gogame.batch_env_step(w, None, rng, 7.5, 'real', True, out=eo))
        im = ev(lambda: gogame._invalid_mask_dev(st, None))
        for k in held: os.environ.pop(k)
        line.append('B %5d: ns %.1f es %.1f im %.1f' % (B, ns, es, im))
    print('%dx%d  ' % (N, N) + ' | '.join(line), flush=True)
