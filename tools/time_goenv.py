"""Config 1: single GoEnv driven through the reference-style API (NumPy in/out), steps per second."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gymgo_amd.envs import make
for size in (7, 19):
    env = make('gym_go:go-v0', size=size, komi=0, reward_method='real')
    env.reset()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 3.0:
        a = env.uniform_random_action()
        state, reward, done, info = env.step(a)
        n += 1
        if done:
            env.reset()
    print('GoEnv %dx%d: %.0f steps/s through step() + uniform_random_action()' % (size, size, n / (time.perf_counter() - t0)))
