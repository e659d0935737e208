"""13x13 entry points for ONE library (LIB=<path relative to the repo root>): what a different row stride of the 13x13 kernels does.
    LIB=ab_tmp/libgg_rs16.so python tools/exp/ab_13.py"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['LIB'])
from gymgo_amd import gogame
N = 13


def timed(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


out = []
h = hashlib.sha1()
for B in (4096, 24576, 32768):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 5, 0, 'cuda')
    gogame.batch_rollout(st, rng, 300, True)
    out.append('rollout x256 B %d %.1f us' % (B, timed(lambda: gogame.batch_rollout(st, rng, 256, True), 8)))
    h.update(st.cpu().numpy().tobytes())
B = 65536
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 6, 0, 'cuda')
gogame.batch_rollout(st, rng, 120, True)
acts = gogame.batch_sample_actions(st, rng)
nxt = torch.empty_like(st)
out.append('next_states %.1f us' % timed(lambda: gogame.batch_next_states(st, acts, check=False, out=nxt)))
from gymgo_amd import state_utils
turn = torch.zeros(B, dtype=torch.int32, device='cuda')
out.append('invalid_mask %.1f us' % timed(lambda: state_utils.batch_compute_invalid_moves(st, None, None)))
out.append('rollout x1 %.1f us' % timed(lambda: gogame.batch_rollout(st, rng, 1, True)))
tr = gogame.batch_track(st)
out.append('track %.1f us' % timed(lambda: gogame.batch_track(st)))
obs = torch.empty_like(st); eo = None
def step():
    global eo
    eo = gogame.batch_env_step_tracked(tr, None, rng, 7.5, 'real', True, out=eo, states_out=obs)
out.append('env step tracked %.1f us' % timed(step))
out.append('rollout_tracked x1 %.1f us' % timed(lambda: gogame.batch_rollout_tracked(tr, rng, 1, True)))
ch = gogame.batch_children(st[:4096])
out.append('children 4096 %.1f us' % timed(lambda: gogame.batch_children(st[:4096], out=ch), 8))
h.update(st.cpu().numpy().tobytes()); h.update(tr.cpu().numpy().tobytes()); h.update(nxt.cpu().numpy().tobytes())
print('%-22s %s | digest %s' % (os.environ.get('LIB', 'shipped'), '  '.join(out), h.hexdigest()[:10]), flush=True)
