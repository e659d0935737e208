"""The per-ply entry points on BYTE PLANES at small batches as hipGraph nodes (us per node) for one library:
gg_batch_next_states (ping-pong of two buffers), gg_batch_env_step (drawn moves), gg_batch_invalid_mask, gg_batch_rollout 1 ply.
LIB=<path relative to the repo root> (default: the shipped one); GGSIZES="9:4096,..."."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['LIB'])
from gymgo_amd import gogame


def graph_us(fn, nodes=64, reps=8):
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side): fn()
    side.synchronize()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(nodes): fn()
    graph.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): graph.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * nodes) * 1e3


sizes = [tuple(int(x) for x in t.split(':')) for t in os.environ.get('GGSIZES', '9:1024,9:4096,13:4096,19:1024,19:4096,19:16384').split(',')]
for N, B in sizes:
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
    ch = max(1, B // 16)
    for g in range(1, 16):
        gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * (8 if N <= 9 else 20 if N <= 13 else 40), True)
    torch.cuda.synchronize()
    acts = gogame.batch_sample_actions(st, rng)
    bufs = [st.clone(), torch.empty_like(st)]
    status = torch.empty(B, dtype=torch.int32, device='cuda')
    k = [0]
    def ns():
        gogame.batch_next_states(bufs[k[0] & 1], acts, check=False, out=bufs[(k[0] + 1) & 1], status=status)
    t_ns = graph_us(ns)
    es = st.clone(); er = rng.clone()
    out = gogame.batch_env_step(es, None, er, 7.5, 'real', True)
    t_es = graph_us(lambda: gogame.batch_env_step(es, None, er, 7.5, 'real', True, out=out))
    mask = gogame.batch_invalid_mask(st) if hasattr(gogame, 'batch_invalid_mask') else None
    r1 = st.clone(); rr = rng.clone()
    t_r1 = graph_us(lambda: gogame.batch_rollout(r1, rr, 1, True))
    torch.cuda.synchronize()
    dg = hashlib.sha1(bufs[1].cpu().numpy().tobytes() + es.cpu().numpy().tobytes() + r1.cpu().numpy().tobytes()).hexdigest()[:8]
    print('%s N %2d B %6d: next_states %.2f us | env_step %.2f us | rollout 1 ply %.2f us | %s' % (os.environ.get('LIB', 'shipped'), N, B, t_ns, t_es, t_r1, dg), flush=True)
