"""A/B build: one-ply launches on TRACKED boards as hipGraph nodes - gg_batch_rollout_tracked (1 / 2 / 4 plies) and
gg_batch_env_step_tracked (with / without the observation) - as single-wave and as four-wave workgroups (GG_AB_WPB = 1 / 4).
LIB=<A/B library>; sizes from GGSIZES="9:4096,13:4096,..."."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, os.environ.get('LIB', 'ab_libs/libgg_ab3.so'))
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


sizes = [tuple(int(x) for x in t.split(':')) for t in os.environ.get('GGSIZES', '9:1024,9:4096,9:8192,9:16384,13:4096,13:8192,19:2048,19:4096').split(',')]
for N, B in sizes:
    row = []
    for wpb in ('1', '4'):
        os.environ['GG_AB_WPB'] = wpb
        st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
        ch = max(1, B // 16)
        for g in range(1, 16):
            gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * (8 if N <= 9 else 20 if N <= 13 else 40), True)
        tr = gogame.batch_track(st)
        res = []
        for F in (1, 2, 4):
            res.append(graph_us(lambda: gogame.batch_rollout_tracked(tr, rng, F, True)))
        out = gogame.batch_env_step_tracked(tr, None, rng, 7.5, 'real', True)
        obs = torch.empty(B, 6, N, N, dtype=torch.uint8, device='cuda')
        res.append(graph_us(lambda: gogame.batch_env_step_tracked(tr, None, rng, 7.5, 'real', True, out=out, states_out=obs)))
        res.append(graph_us(lambda: gogame.batch_env_step_tracked(tr, None, rng, 7.5, 'real', True, out=out)))
        torch.cuda.synchronize()
        row.append((res, hashlib.sha1(tr.cpu().numpy().tobytes() + obs.cpu().numpy().tobytes()).hexdigest()[:8]))
    (a, da), (b, db) = row
    print('N %2d B %6d  us per node, 1-wave -> 4-wave workgroups: rollout 1 ply %.2f -> %.2f | 2 plies %.2f -> %.2f | 4 plies %.2f -> %.2f | env step + obs %.2f -> %.2f | env step %.2f -> %.2f | %s'
          % (N, B, a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], 'same digest' if da == db else 'DIGESTS DIFFER'), flush=True)
