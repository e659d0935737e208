"""`state_utils`-compatible entry points (mirrors gym_go/state_utils.py:24-250).

compute_invalid_moves / batch_compute_invalid_moves run the HIP liberty analysis
(gg_batch_invalid_mask); update_pieces / batch_update_pieces (capture resolution,
gym_go/state_utils.py:159-211) run gg_batch_update_pieces (inside gg_batch_next_states the same
step is fused, exactly where the reference calls it: gym_go/gogame.py:68, :127-128).
"""
import numpy as np
import torch

from gymgo_amd import _lib, govars
from gymgo_amd.gogame import _Box, _invalid_mask_dev

neighbor_deltas = np.array([[-1, 0], [1, 0], [0, -1], [0, 1]])  # gym_go/state_utils.py:21

# The reference's scipy.ndimage structuring elements (gym_go/state_utils.py:7-19), kept as module constants for callers
# that import them; the kernels encode the same 4-connectivity in their shifts.
surround_struct = np.zeros((3, 3), dtype=np.int64)
surround_struct[tuple((neighbor_deltas + 1).T)] = 1          # the four neighbours of the centre
group_struct = np.zeros((3, 3, 3), dtype=np.int64)
group_struct[1] = surround_struct
group_struct[1, 1, 1] = 1                                    # 4-connected inside a board, boards not connected


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


def _killed_groups(mask):
    """0/1 mask [N, N] -> list of [k, 2] coordinate arrays, one per 4-connected group, groups and points in
    raster order (the order scipy.ndimage.label / np.argwhere give the reference)."""
    mask = np.asarray(mask).astype(bool)
    n = mask.shape[0]
    seen = np.zeros_like(mask)
    groups = []
    for r0, c0 in np.argwhere(mask):
        if seen[r0, c0]:
            continue
        stack, cells = [(r0, c0)], []
        seen[r0, c0] = True
        while stack:
            r, c = stack.pop()
            cells.append((r, c))
            for dr, dc in neighbor_deltas:
                rr, cc = r + dr, c + dc
                if 0 <= rr < n and 0 <= cc < n and mask[rr, cc] and not seen[rr, cc]:
                    seen[rr, cc] = True
                    stack.append((rr, cc))
        groups.append(np.array(sorted(cells)))
    return groups


def _adj_table(batch_adj_locs, n):
    """list of [k, 2] location arrays -> int32 [B, K] flat indices, padded with -1 (K = the longest list, >= 4)."""
    rows = [np.asarray(a, dtype=np.int64).reshape(-1, 2) for a in batch_adj_locs]
    K = max([4] + [len(r) for r in rows])
    table = np.full((len(rows), K), -1, dtype=np.int32)
    for i, r in enumerate(rows):
        ok = ((r >= 0) & (r < n)).all(axis=1)
        table[i, :len(r)] = np.where(ok, r[:, 0] * n + r[:, 1], -1)
    return table


def batch_update_pieces(batch_non_pass, batch_state, batch_adj_locs, batch_player):
    """gym_go/state_utils.py:183-211: removes captured opponent groups IN PLACE for the games listed in
    batch_non_pass and returns their killed groups (list per game of [k, 2] arrays).  Each game is treated
    with update_pieces semantics (the reference's zip mis-alignment with passes, :187-193, is not reproduced).
    The adj_locs go to the kernel as they are (gg_batch_update_pieces): nothing is inferred about the position."""
    is_t = isinstance(batch_state, torch.Tensor)
    host = batch_state if is_t else np.asarray(batch_state)
    idx = np.asarray(batch_non_pass, dtype=np.int64).reshape(-1)
    players = np.asarray(batch_player, dtype=np.int32).reshape(-1)
    n = host.shape[-1]
    adj = _adj_table(batch_adj_locs, n)
    sub = _Box(host[idx] if not is_t else batch_state[torch.as_tensor(idx, device=batch_state.device)])
    t = sub.t.clone()
    B, _, N, _ = t.shape
    killed = torch.empty((B, N, N), dtype=torch.uint8, device=t.device)
    adj_t = torch.from_numpy(adj).to(t.device)
    pls = torch.from_numpy(players).to(t.device)
    code = _lib.lib().gg_batch_update_pieces(_lib.dev_ptr(t, torch.uint8, 'states'), _lib.dev_ptr(adj_t, torch.int32, 'adj'),
                                             adj.shape[1], _lib.dev_ptr(pls, torch.int32, 'players'),
                                             _lib.dev_ptr(killed, torch.uint8, 'killed'), B, N, _lib.stream_ptr(t.device))
    _lib.check(code, 'gg_batch_update_pieces')
    if is_t:
        batch_state[torch.as_tensor(idx, device=batch_state.device)] = t.to(batch_state.dtype)
    else:
        batch_state[idx] = t.cpu().numpy().astype(batch_state.dtype)
    return [_killed_groups(k) for k in killed.cpu().numpy()]


def update_pieces(state, adj_locs, player):
    """gym_go/state_utils.py:159-180: mutates `state`, returns the list of killed groups."""
    batch = state[None] if isinstance(state, torch.Tensor) else np.asarray(state)[None]
    out = batch_update_pieces([0], batch, [adj_locs], [player])[0]
    if not isinstance(state, torch.Tensor):
        state[...] = batch[0]
    return out
