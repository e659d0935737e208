"""Kernel-rate timing of the per-ply entry points through the raw C-ABI (preallocated buffers, no Python-side
allocation between launches), 19x19 x 65 536 mid-game boards.  Prints one JSON object."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gymgo_amd import _lib, gogame

B, N = 65536, 19
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 3)
for g in range(16):
    gogame.batch_rollout(st[g * 4096:(g + 1) * 4096], rng[g * 4096:(g + 1) * 4096], 150 + 20 * g, True)
acts = gogame.batch_sample_actions(st, rng)
out = torch.empty_like(st); status = torch.empty(B, dtype=torch.int32, device='cuda')
rew = torch.empty(B, dtype=torch.float32, device='cuda'); dones = torch.empty(B, dtype=torch.uint8, device='cuda')
taken = torch.empty(B, dtype=torch.int32, device='cuda')
pk = gogame.batch_pack(st); pko = torch.empty_like(pk)
L = _lib.lib(); s = _lib.stream_ptr(st.device)
P = lambda t: _lib.dev_ptr(t, t.dtype, 'x')

def timed(fn, reps=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

res = {}
def add(name, fn, bytes_per_board=None):
    t = timed(fn)
    res[name] = {'us_per_launch': round(t * 1e6, 2), 'steps_per_s': round(B / t, 1)}
    if bytes_per_board:
        res[name]['roofline_frac'] = round(B * bytes_per_board / t / 8e12, 4)
S = 6 * N * N
add('gg_batch_next_states', lambda: L.gg_batch_next_states(P(st), P(acts), P(out), P(status), B, N, 0, s), 2 * S + 4)
add('gg_batch_rollout_1ply', lambda: L.gg_batch_rollout(P(st), P(rng), None, None, B, N, 1, 1, s), 2 * S + 4)
add('gg_batch_env_step_real', lambda: L.gg_batch_env_step(P(st), None, P(rng), P(rew), P(dones), P(status), P(taken), B, N, 7.5, 0, 1, s), 2 * S + 4)
add('gg_batch_env_step_heuristic', lambda: L.gg_batch_env_step(P(st), None, P(rng), P(rew), P(dones), P(status), P(taken), B, N, 7.5, 1, 1, s), 2 * S + 4)
add('gg_batch_next_states_packed', lambda: L.gg_batch_next_states_packed(P(pk), P(acts), P(pko), P(status), B, N, 0, s))
add('gg_batch_env_step_packed_real', lambda: L.gg_batch_env_step_packed(P(pk), None, P(rng), P(rew), P(dones), P(status), P(taken), B, N, 7.5, 0, 1, s))
add('gg_batch_rollout_packed_1ply', lambda: L.gg_batch_rollout_packed(P(pk), P(rng), None, None, B, N, 1, 1, s))
print(json.dumps(res, indent=1))
