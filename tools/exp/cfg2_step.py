import sys, time; sys.path.insert(0, '/root/repo')
import torch
from gymgo_amd.envs import GoVecEnv, GoVecEnvParts
for layout in ('tracked', 'bytes', 'packed'):
    env = GoVecEnv(4096, 9, layout=layout); env.rollout(30)
    env.step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): env.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 500
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): env.step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(64): env.step()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); dg = (time.perf_counter() - t0) / 20 / 64
    print('%-8s GoVecEnv(4096, 9).step: loop %.2f us (%.3e steps/s), hipGraph of 64: %.2f us (%.3e steps/s)' % (layout, dt * 1e6, 4096 / dt, dg * 1e6, 4096 / dg))
