#!/usr/bin/env python
"""Launches with KNOWN HBM byte counts, for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this box
(MI355X_MICROARCH.md: FETCH_SIZE halves wide coalesced reads on gfx950, WRITE_SIZE is uncalibrated):
  gg_batch_unpack_states  65 536 packed 19x19 boards -> byte planes: reads 232 B, writes 2 166 B per board with the same
                          store pattern as the rollout kernel's write-back (aligned 16-byte vectors + ragged byte stores)
  gg_batch_pack_states    the reverse: reads 1 444 B (planes 0, 1, 3 + flag bytes), writes 232 B per board
Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (tools/profile_round.sh); tools/summarize_profiles.py
divides the counters by these byte counts."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymgo_amd import gogame  # noqa: E402

B, N = 65536, 19
st = gogame.batch_init_state(B, N, device='cuda')
rng = gogame.rng_seed(B, 3)
gogame.batch_rollout(st, rng, 200, True)
pk = gogame.batch_pack(st)
for _ in range(8):
    st2 = gogame.batch_unpack(pk, N)
    pk2 = gogame.batch_pack(st2)
torch.cuda.synchronize()
assert torch.equal(st2, st) and torch.equal(pk2, pk)
# config 5's launch under the same counters: 8 192 parents (the stationary mix) x 362 slots, 786 258 B per parent by the
# algorithm (1 444 B read, 362 x 2 166 B written) - tools/summarize_profiles.py sets the counters against that
kids = torch.empty((8192, N * N + 1, 6, N, N), dtype=torch.uint8, device='cuda')
for _ in range(4):
    gogame.batch_children(st[:8192], out=kids)
torch.cuda.synchronize()
print('calib ok')
