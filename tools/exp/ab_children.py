"""gg_batch_children (config 5: 8 192 parents of 19x19) padded and un-padded, by game phase, through the bench line's own
harness (bench.children_record) for ONE library (LIB=<path relative to the repo root>, default: the shipped one)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gymgo_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['LIB'])
from gymgo_amd import gogame
import bench
N, B = 19, 8192
st = gogame.batch_init_state(B, N, device='cuda'); rng = gogame.rng_seed(B, 77)
for g in range(1, 16):
    gogame.batch_rollout(st[g * 512:(g + 1) * 512], rng[g * 512:(g + 1) * 512], g * 40, True)
gogame.batch_rollout(st, rng, 300, True)
rec = bench.children_record(torch, torch.device('cuda', 0), st, 'stationary mix', by_phase=True, reps=8)
print(os.environ.get('LIB', 'shipped'), json.dumps(rec, indent=1), flush=True)
