"""The shapes the one-row-per-lane kernels (gg_lat.h) serve, for a LIST of libraries (paths relative to the repo root), interleaved,
with state digests: fused byte-plane rollouts (config 2 and neighbours), tracked rollouts (256 plies / one ply as a hipGraph node)
and the tracked env step.
    python tools/exp/lat_libs.py ab_tmp/base.so gymgo_amd/libgymgo_amd.so"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] != 'run':
    for rep in range(2):
        for lib in sys.argv[1:]:
            r = subprocess.run([sys.executable, __file__, 'run', lib], capture_output=True, text=True)
            print(lib + '\n  ' + (r.stdout.strip().replace(' | ', '\n  ') or r.stderr[-600:]), flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch, hashlib
from gymgo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, sys.argv[2])
from gymgo_amd import gogame


def ev(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def graph_us(fn, nodes=32, reps=8):
    side = torch.cuda.Stream(); graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
    side.synchronize()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(nodes):
            fn()
    graph.replay(); torch.cuda.synchronize()
    return ev(graph.replay, reps) / nodes * 1e3


out = []
for N, B in ((9, 4096), (9, 16384), (13, 4096), (19, 2048), (19, 4096)):
    st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
    ch = B // 16
    for g in range(1, 16):
        gogame.batch_rollout(st[g * ch:(g + 1) * ch], rng[g * ch:(g + 1) * ch], g * (8 if N <= 9 else 20 if N <= 13 else 40), True)
    gogame.batch_rollout(st, rng, 256, True)
    ms = ev(lambda: gogame.batch_rollout(st, rng, 256, True), 12)
    tr = gogame.batch_track(st); rg = rng.clone()
    mt = ev(lambda: gogame.batch_rollout_tracked(tr, rg, 256, True), 12)
    g1 = graph_us(lambda: gogame.batch_rollout_tracked(tr, rg, 1, True))
    obs = torch.empty_like(st)
    eo = (torch.empty(B, dtype=torch.float32, device='cuda'), torch.empty(B, dtype=torch.uint8, device='cuda'),
          torch.empty(B, dtype=torch.int32, device='cuda'), torch.empty(B, dtype=torch.int32, device='cuda'))
    ge = graph_us(lambda: gogame.batch_env_step_tracked(tr, None, rg, 7.5, 'real', True, out=eo, states_out=obs))
    dg = hashlib.sha1(st.cpu().numpy().tobytes() + tr.cpu().numpy().tobytes()).hexdigest()[:8]
    out.append('%dx%d B%d: fused %.4f ms (%.3e/s)  tracked fused %.4f ms (%.3e/s)  tracked 1-ply node %.2f us  env-step node %.2f us  %s'
               % (N, N, B, ms, B * 256 / ms * 1e3, mt, B * 256 / mt * 1e3, g1, ge, dg))
print(' | '.join(out))
