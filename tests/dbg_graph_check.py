"""hipGraph capture of a per-ply rollout loop through torch.cuda.CUDAGraph: the C-ABI launches go to torch's capture
stream, so a whole K-ply loop of 1-ply launches replays as one graph (no host launch overhead)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gymgo_amd import gogame
from oracle import c_oracle
B, N, K = 65536, 19, 32
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 5)
gogame.batch_rollout(st, rng, 200, True)
ref_st, ref_rng = st.clone(), rng.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): gogame.batch_rollout(st, rng, 1, True)       # warm-up on the side stream
torch.cuda.current_stream().wait_stream(s)
st.copy_(ref_st); rng.copy_(ref_rng)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(K): gogame.batch_rollout(st, rng, 1, True)
st.copy_(ref_st); rng.copy_(ref_rng)
g.replay(); torch.cuda.synchronize()
want = ref_st.clone(); wr = ref_rng.clone()
gogame.batch_rollout(want, wr, K, True)                              # the same K plies fused
assert torch.equal(st, want) and torch.equal(rng, wr), 'graph replay != fused rollout'
def rate(fn, reps, plies):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return B * plies * reps / (e0.elapsed_time(e1) * 1e-3)
print('eager 1-ply launches : %.3e steps/s' % rate(lambda: gogame.batch_rollout(st, rng, 1, True), 64, 1))
print('hipGraph of %d launches: %.3e steps/s' % (K, rate(lambda: g.replay(), 8, K)))
print('fused %d plies/launch  : %.3e steps/s' % (K, rate(lambda: gogame.batch_rollout(st, rng, K, True), 8, K)))
print('graph replay bit-exact vs fused: ok')
