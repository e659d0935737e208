"""The tracked env step as K independent sub-batches on K HIP streams (each sub-batch is its own env: no cross-stream
dependency): does one sub-batch's write-back overlap another's ply?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import gogame
B, N, STEPS = 65536, 19, 60
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 3)
for g in range(16):
    gogame.batch_rollout(st[g*4096:(g+1)*4096], rng[g*4096:(g+1)*4096], 100 + 40*g, True)
for K in (1, 2, 4, 8):
    per = B // K
    streams = [torch.cuda.Stream() for _ in range(K)]
    parts = []
    for k in range(K):
        s = st[k*per:(k+1)*per].contiguous()
        parts.append(dict(tr=gogame.batch_track(s), rng=rng[k*per:(k+1)*per].clone(), obs=torch.empty_like(s),
                          out=(torch.empty(per, dtype=torch.float32, device='cuda'), torch.empty(per, dtype=torch.uint8, device='cuda'),
                               torch.empty(per, dtype=torch.int32, device='cuda'), torch.empty(per, dtype=torch.int32, device='cuda'))))
    torch.cuda.synchronize()
    def run(n):
        for _ in range(n):
            for k in range(K):
                with torch.cuda.stream(streams[k]):
                    p = parts[k]
                    gogame.batch_env_step_tracked(p['tr'], None, p['rng'], 7.5, 'real', True, out=p['out'], states_out=p['obs'])
    run(5); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for sx in streams: sx.wait_stream(torch.cuda.current_stream())
    run(STEPS)
    for sx in streams: torch.cuda.current_stream().wait_stream(sx)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / STEPS * 1e3
    print('K = %d streams x %6d games: %.1f us per step of all %d games = %.3e env steps/s' % (K, per, us, B, B / us * 1e6), flush=True)
