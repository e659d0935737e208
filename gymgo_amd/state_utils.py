"""`state_utils`-compatible entry points (mirrors gym_go/state_utils.py:24-250).

compute_invalid_moves / batch_compute_invalid_moves run the HIP liberty analysis
(gg_batch_invalid_mask).  update_pieces / batch_update_pieces (capture resolution,
gym_go/state_utils.py:159-211) have no stand-alone device entry: they are fused into
gg_batch_next_states, exactly where the reference calls them (gym_go/gogame.py:68, :127-128).
"""
import numpy as np
import torch

from gymgo_amd import govars
from gymgo_amd.gogame import _Box, _invalid_mask_dev

neighbor_deltas = np.array([[-1, 0], [1, 0], [0, -1], [0, 1]])  # gym_go/state_utils.py:21


def _ko_tensor(batch_ko, B, N, device):
    if batch_ko is None:
        return None
    if isinstance(batch_ko, torch.Tensor):
        return batch_ko.to(device=device, dtype=torch.int32).contiguous()
    flat = np.full(B, -1, dtype=np.int32)
    for i, k in enumerate(batch_ko):
        if k is not None:
            k = np.asarray(k)
            flat[i] = int(k) if k.ndim == 0 else int(k[0]) * N + int(k[1])
    return torch.from_numpy(flat).to(device)


def batch_compute_invalid_moves(batch_state, batch_player, batch_ko_protect):
    """gym_go/state_utils.py:86-156: invalid moves for the OPPONENT of batch_player (the side that
    just moved).  batch_player=None takes it from the turn plane (player = 1 - turn).
    batch_ko_protect: None, a list of None / (r, c), or an int tensor of flat indices (-1 = none)."""
    box = _Box(batch_state)
    t = box.t
    B, _, N, _ = t.shape
    if batch_player is not None:
        p = torch.as_tensor(np.asarray(batch_player) if not isinstance(batch_player, torch.Tensor) else batch_player)
        p = p.to(device=t.device, dtype=torch.uint8).reshape(B)
        t = t.clone()
        t[:, govars.TURN_CHNL] = (1 - p)[:, None, None]
    mask = _invalid_mask_dev(t, _ko_tensor(batch_ko_protect, B, N, t.device))
    if box.numpy:
        return mask.cpu().numpy() > 0
    return mask


def compute_invalid_moves(state, player, ko_protect=None):
    """gym_go/state_utils.py:24-83."""
    if isinstance(state, torch.Tensor):
        return batch_compute_invalid_moves(state[None], [player], [ko_protect])[0]
    return batch_compute_invalid_moves(np.asarray(state)[None], [player], [ko_protect])[0]


def adj_data(state, action2d, player):
    """gym_go/state_utils.py:214-223: on-board neighbours of a point; surrounded = all hold an opponent stone."""
    s = state.cpu().numpy() if isinstance(state, torch.Tensor) else np.asarray(state)
    nbrs = neighbor_deltas + np.asarray(action2d)
    nbrs = nbrs[((nbrs >= 0) & (nbrs < s.shape[1])).all(axis=1)]
    return nbrs, bool((s[1 - player][nbrs[:, 0], nbrs[:, 1]] > 0).all())


def batch_adj_data(batch_state, batch_action2d, batch_player):
    """gym_go/state_utils.py:226-232."""
    out = [adj_data(s, a, p) for s, a, p in zip(batch_state, batch_action2d, batch_player)]
    return [o[0] for o in out], [o[1] for o in out]


def set_turn(state):
    """gym_go/state_utils.py:235-241: flip the turn plane IN PLACE."""
    state[govars.TURN_CHNL] = 1 - state[govars.TURN_CHNL]


def batch_set_turn(batch_state):
    """gym_go/state_utils.py:244-250."""
    batch_state[:, govars.TURN_CHNL] = 1 - batch_state[:, govars.TURN_CHNL]
