import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gymgo_amd import gogame, _lib
B, N = 65536, 19
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 3)
for g in range(16):
    gogame.batch_rollout(st[g*4096:(g+1)*4096], rng[g*4096:(g+1)*4096], 100 + 40*g, True)
acts = gogame.batch_sample_actions(st, rng)
out = torch.empty_like(st); status = torch.empty(B, dtype=torch.int32, device='cuda')
def ev(fn, reps):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
f = lambda: gogame.batch_next_states(st, acts, check=False, out=out, status=status)
print('api out= us/call', ev(f, 64))
L = _lib.lib(); sp = torch.cuda.current_stream().cuda_stream
a0, a1, a2, a3 = st.data_ptr(), acts.data_ptr(), out.data_ptr(), status.data_ptr()
g = lambda: L.gg_batch_next_states(a0, a1, a2, a3, B, N, 0, sp)
print('raw ctypes us/call', ev(g, 64))
# host cost alone: tiny batch
st1 = st[:2].clone(); o1 = torch.empty_like(st1); s1 = status[:2]; ac1 = acts[:2].clone()
t0 = time.perf_counter()
for _ in range(2000): gogame.batch_next_states(st1, ac1, check=False, out=o1, status=s1)
torch.cuda.synchronize(); print('host us/call (B=2, api)', (time.perf_counter() - t0) / 2000 * 1e6)
b0, b1, b2, b3 = st1.data_ptr(), ac1.data_ptr(), o1.data_ptr(), s1.data_ptr()
t0 = time.perf_counter()
for _ in range(2000): L.gg_batch_next_states(b0, b1, b2, b3, 2, N, 0, sp)
torch.cuda.synchronize(); print('host us/call (B=2, raw)', (time.perf_counter() - t0) / 2000 * 1e6)
# graph
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): f()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(32): f()
print('graph us/call', ev(lambda: gr.replay(), 8) / 32)
# alternate in/out (ping-pong) to see cache effects
out2 = torch.empty_like(st)
def pp():
    gogame.batch_next_states(st, acts, check=False, out=out, status=status)
    gogame.batch_next_states(out, acts, check=False, out=out2, status=status)
print('pingpong us/call', ev(pp, 32) / 2)
