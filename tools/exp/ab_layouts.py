import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gymgo_amd import gogame
B, N, F = 65536, 19, 256
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 20260927)
ch = B // 16
for g in range(16):
    gogame.batch_rollout(st[g*ch:(g+1)*ch], rng[g*ch:(g+1)*ch], g * 40, True)
def ev(f, n=8):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for rep in range(2):
    s1, r1 = st.clone(), rng.clone()
    la = torch.full((B,), -1, dtype=torch.int32, device='cuda'); sd = torch.zeros(B, dtype=torch.int64, device='cuda')
    print('bytes  + last_actions + steps_done: %.4f ms' % ev(lambda: gogame.batch_rollout(s1, r1, F, True, la, sd)))
    s1, r1 = st.clone(), rng.clone()
    print('bytes  plain:                       %.4f ms' % ev(lambda: gogame.batch_rollout(s1, r1, F, True)))
    t1, r1 = gogame.batch_track(st), rng.clone()
    print('tracked plain:                      %.4f ms' % ev(lambda: gogame.batch_rollout_tracked(t1, r1, F, True)))
    t1, r1 = gogame.batch_track(st), rng.clone()
    print('tracked + last_actions + steps_done:%.4f ms' % ev(lambda: gogame.batch_rollout_tracked(t1, r1, F, True, la, sd)))
    p1, r1 = gogame.batch_pack(st), rng.clone()
    print('packed plain:                       %.4f ms' % ev(lambda: gogame.batch_rollout_packed(p1, r1, F, True)))
