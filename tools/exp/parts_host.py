"""Host cost of GoVecEnvParts.step_part against the bare calls it is made of (small batches: the launches are short, the
loop is host-bound) and the full-size two-half step.  python tools/exp/parts_host.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import gogame, _lib
from gymgo_amd.envs import GoVecEnv, GoVecEnvParts
dev = torch.device('cuda')


def rate(f, n=2000):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6, (time.perf_counter() - t0) / n * 1e6


env = GoVecEnv(1024, 19)
print('GoVecEnv.step (1024 games)            host %.1f us, with GPU %.1f us' % rate(lambda: env.step()))
print('gogame.batch_env_step_tracked          host %.1f us, with GPU %.1f us' % rate(lambda: gogame.batch_env_step_tracked(
    env.tracked, None, env.rng, 0.0, 'real', True, out=env._step_out, states_out=env._obs, steps_done=env.steps_done)))
s = torch.cuda.Stream()
ev = torch.cuda.Event()
cur = torch.cuda.current_stream()
print('s.wait_stream(cur)                     host %.1f us' % rate(lambda: s.wait_stream(cur))[0])
print('ev.record(cur); s.wait_event(ev)       host %.1f us' % rate(lambda: (ev.record(cur), s.wait_event(ev)))[0])
print('torch.cuda.current_stream()            host %.1f us' % rate(lambda: torch.cuda.current_stream(dev))[0])
def ctx():
    with torch.cuda.stream(s):
        pass
print('with torch.cuda.stream(s): pass        host %.1f us' % rate(ctx)[0])
print('ev.record(s)                           host %.1f us' % rate(lambda: ev.record(s))[0])
p = GoVecEnvParts(2048, 19, parts=2)
print('GoVecEnvParts.step_part (1024 games)   host %.1f us, with GPU %.1f us' % rate(lambda: p.step_part(0)))
for B in (65536,):
    p = GoVecEnvParts(B, 19, parts=2)
    for h in range(2):
        p.rollout_part(h, 100)
    def both():
        p.step_part(0); p.step_part(1)
    h_us, g_us = rate(both, 200)
    print('two halves of %d games: host %.1f us per full step, with GPU %.1f us -> %.3e steps/s' % (B, h_us, g_us, B / g_us * 1e6))
    e1 = GoVecEnv(B, 19); e1.rollout(100)
    h_us, g_us = rate(lambda: e1.step(), 200)
    print('one env of %d games:    host %.1f us per step, with GPU %.1f us -> %.3e steps/s' % (B, h_us, g_us, B / g_us * 1e6))
for parts in (2, 3, 4):
    p = GoVecEnvParts(65536, 19, parts=parts)
    for h in range(parts):
        p.rollout_part(h, 100)
    def free():
        for h in range(parts):
            p.step_part(h)
    def pingpong():
        for h in range(parts):
            p.wait(h); p.step_part(h)
    for name, f in (('free-running', free), ('wait(h) before every step_part(h)', pingpong)):
        h_us, g_us = rate(f, 200)
        print('%d parts, %-34s host %.1f us per full step, with GPU %.1f us -> %.3e steps/s' % (parts, name, h_us, g_us, 65536 / g_us * 1e6))
# policy-weighted steps (bfloat16 / float32 weights): one env against two free-running parts
for dt in (torch.bfloat16, torch.float32):
    B = 65536
    w = (torch.rand((B, 362), device='cuda') ** 4).to(dt)
    e1 = GoVecEnv(B, 19); e1.rollout(100)
    h_us, g1 = rate(lambda: e1.step(probs=w), 200)
    h_us, g0 = rate(lambda: e1.step(), 200)
    p = GoVecEnvParts(B, 19, parts=2)
    for h in range(2):
        p.rollout_part(h, 100)
    ws = [w[lo:hi] for lo, hi in p.bounds]
    def both_w():
        p.step_part(0, probs=ws[0]); p.step_part(1, probs=ws[1])
    def both_u():
        p.step_part(0); p.step_part(1)
    h_us, g2 = rate(both_w, 200)
    h_us, g3 = rate(both_u, 200)
    print('%s weights: one env %.1f us (uniform %.1f: x%.2f); two parts %.1f us (uniform %.1f: x%.2f)' % (dt, g1, g0, g0 / g1, g2, g3, g3 / g2))
