"""Host time of GoVecEnv.step() (tracked layout) by Python function: cProfile over a few thousand steps of a tiny batch."""
import os, sys, cProfile, pstats, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd.envs import GoVecEnv
env = GoVecEnv(256, 9, komi=7.5, reward_method='real', layout='tracked')
for _ in range(200): env.step()
torch.cuda.synchronize()
n = 20000
t0 = time.perf_counter()
for _ in range(n): env.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('env.step(): %.2f us per call on the host (%.2f incl. the final sync)' % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
acts = env.sample_actions() if hasattr(env, 'sample_actions') else None
pr = cProfile.Profile()
pr.enable()
for _ in range(5000): env.step()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
