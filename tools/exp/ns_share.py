"""gg_batch_next_states: the uneven split of a SIMD's pairs among its three waves (A/B build: GG_AB_NS_S1 / S2 / EVEN)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    cfgs = [('EVEN', '', '')]
    for n0, n1 in ((14, 10), (15, 10), (14, 11), (15, 9), (16, 9), (15, 11), (16, 10), (13, 11), (17, 9)):
        cfgs.append(('', '%.4f' % ((n0 + 0.5) / 32), '%.4f' % ((n0 + n1 + 0.5) / 32)))
    for cfg in cfgs:
        env = dict(os.environ, LIB='libgymgo_ab.so')
        if cfg[0]: env['GG_AB_NS_EVEN'] = '1'
        else: env['GG_AB_NS_S1'], env['GG_AB_NS_S2'] = cfg[1], cfg[2]
        print(cfg, subprocess.run([sys.executable, __file__, 'run'], env=env, capture_output=True, text=True).stdout.strip(), flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', os.environ['LIB'])
from gymgo_amd import gogame
N = 19
for B in (65536,):
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
    print('B %d: %.1f us' % (B, a.elapsed_time(b) / 32 * 1e3), end='   ')
