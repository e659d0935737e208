"""`gogame`-compatible function library over the HIP kernels (mirrors gym_go/gogame.py:22-468:
same names, positional order, return shapes and error behaviour).

Containers
  * torch uint8 device tensors ([6,N,N] / [B,6,N,N]) are the native form: results are device tensors.
  * NumPy arrays (the reference's float64 states) are accepted everywhere for drop-in use: they are
    moved to the device, run through the same kernels, and the result comes back as a NumPy array of
    the input dtype.  Nothing is ever computed on the CPU.
Errors: an illegal move raises AssertionError (gym_go/gogame.py:59, :117); action_size() without
arguments raises RuntimeError (:196).
"""
import numpy as np
import torch

from gymgo_amd import _lib, govars

_U8, _I32, _I64 = torch.uint8, torch.int32, torch.int64


def _device():
    if not torch.cuda.is_available():
        raise _lib.GymGoNativeError('no ROCm device visible; gymgo_amd has no CPU path')
    return torch.device('cuda', torch.cuda.current_device())


class _Box:
    """Remembers how the caller passed a state so results go back in the same form."""

    def __init__(self, x):
        self.numpy = not isinstance(x, torch.Tensor)
        if self.numpy:
            x = np.asarray(x)
            self.np_dtype = x.dtype if x.dtype != np.bool_ else np.dtype(np.uint8)
            self.t = torch.from_numpy(np.ascontiguousarray(x).astype(np.uint8, copy=False)).to(_device())
        else:
            if not x.is_cuda:
                raise _lib.GymGoNativeError('state tensors must live on the ROCm device')
            self.t = x.to(_U8).contiguous()

    def back(self, t, dtype=None):
        if self.numpy:
            return t.cpu().numpy().astype(dtype or self.np_dtype)
        return t


def _actions_tensor(actions, B, device):
    if isinstance(actions, torch.Tensor):
        a = actions.to(device=device, dtype=_I32).contiguous()
    else:
        a = torch.as_tensor(np.asarray(actions).astype(np.int32), device=device)
    if a.numel() != B:
        raise ValueError('need one action per state (%d != %d)' % (a.numel(), B))
    return a.reshape(B)


# ------------------------------------------------------------------ raw device calls

def next_states_workspace(batch_size, board_size, device=None):
    """A zero-filled workspace for batch_next_states(..., workspace=): int32 [B, 5N+1] on the device."""
    return torch.zeros((batch_size, tracked_words(board_size)), dtype=_I32, device=device or _device())


def _next_states_dev(states, actions, canonical, out=None, status=None, workspace=None):
    B, C, N, _ = states.shape
    if out is None:
        out = torch.empty_like(states)
    elif out.shape != states.shape:
        raise ValueError('out must have the shape of the states %s' % (tuple(states.shape),))
    if status is None:
        status = torch.empty(B, dtype=_I32, device=states.device)
    if workspace is not None:
        if tuple(workspace.shape) != (B, 5 * N + 1):
            raise ValueError('workspace must be int32 [B, 5N+1] (gogame.next_states_workspace)')
        code = _lib.lib().gg_batch_next_states_ws(
            _lib.dev_ptr(states, _U8, 'states'), _lib.dev_ptr(actions, _I32, 'actions'), _lib.dev_ptr(out, _U8, 'out'),
            _lib.dev_ptr(status, _I32, 'status'), _lib.dev_ptr(workspace, _I32, 'workspace'), B, N, int(bool(canonical)),
            _lib.stream_ptr(states.device))
        _lib.check(code, 'gg_batch_next_states_ws')
        return out, status
    code = _lib.lib().gg_batch_next_states(
        _lib.dev_ptr(states, _U8, 'states'), _lib.dev_ptr(actions, _I32, 'actions'), _lib.dev_ptr(out, _U8, 'out'),
        _lib.dev_ptr(status, _I32, 'status'), B, N, int(bool(canonical)), _lib.stream_ptr(states.device))
    _lib.check(code, 'gg_batch_next_states')
    return out, status


def _children_dev(states, canonical, out=None):
    B, C, N, _ = states.shape
    if out is None:
        out = torch.empty((B, N * N + 1, C, N, N), dtype=_U8, device=states.device)
    elif tuple(out.shape) != (B, N * N + 1, C, N, N):
        raise ValueError('out must be [B, N*N+1, 6, N, N]')
    code = _lib.lib().gg_batch_children(_lib.dev_ptr(states, _U8, 'states'), _lib.dev_ptr(out, _U8, 'children'),
                                        B, N, int(bool(canonical)), _lib.stream_ptr(states.device))
    _lib.check(code, 'gg_batch_children')
    return out


def _children_offsets_dev(states, offsets=None, order=None):
    """(offsets int32 [B+1], order int32 [B]) of gg_batch_children_offsets: exclusive prefix sums of the number of children
    valid_moves() keeps per state, and the states by falling count (the launch order of the expansion)."""
    B, C, N, _ = states.shape
    if offsets is None:
        offsets = torch.empty(B + 1, dtype=_I32, device=states.device)
    elif offsets.numel() < B + 1:          # the launch writes offsets[0 .. B]
        raise ValueError('offsets must hold at least B + 1 = %d int32 (got %d)' % (B + 1, offsets.numel()))
    if order is None:
        order = torch.empty(max(B, 1), dtype=_I32, device=states.device)
    elif order.numel() < B:
        raise ValueError('order must hold at least B = %d int32 (got %d)' % (B, order.numel()))
    code = _lib.lib().gg_batch_children_offsets(_lib.dev_ptr(states, _U8, 'states'), _lib.dev_ptr(offsets, _I32, 'offsets'),
                                                _lib.dev_ptr(order, _I32, 'order'), B, N, _lib.stream_ptr(states.device))
    _lib.check(code, 'gg_batch_children_offsets')
    return offsets, order


def _children_compact_dev(states, canonical, offsets=None, out=None):
    """The un-padded children of every state, concatenated (gg_batch_children_compact) -> (children uint8 [total, 6, N, N],
    offsets int32 [B+1]).  `out`: a caller-owned buffer of at least offsets[B] boards; with the upper bound
    B * (N*N+1) no device -> host read of the total is needed and the launches can be captured in a graph, a smaller buffer
    is checked against the total (one host read) and refused when it is too short."""
    B, C, N, _ = states.shape
    offsets, order = _children_offsets_dev(states, offsets)
    if out is None:
        total = int(offsets[B].item())            # (the one host read of the un-padded form: the size of its result)
        out = torch.empty((total, C, N, N), dtype=_U8, device=states.device)
    elif out.dim() != 4 or tuple(out.shape[1:]) != (C, N, N):
        raise ValueError('out must be [n >= total children, 6, N, N]')
    elif out.shape[0] < B * (N * N + 1):
        # smaller than the upper bound: the true total has to be read back once, or the launch could write past the end
        total = int(offsets[B].item())
        if out.shape[0] < total:
            raise ValueError('out holds %d boards, the un-padded children of this batch are %d' % (out.shape[0], total))
    code = _lib.lib().gg_batch_children_compact(_lib.dev_ptr(states, _U8, 'states'), _lib.dev_ptr(offsets, _I32, 'offsets'),
                                                _lib.dev_ptr(order, _I32, 'order'), _lib.dev_ptr(out, _U8, 'children'), B, N,
                                                int(bool(canonical)), _lib.stream_ptr(states.device))
    _lib.check(code, 'gg_batch_children_compact')
    return out, offsets


def _areas_dev(states, out=None):
    B, C, N, _ = states.shape
    if out is None:
        black = torch.empty(B, dtype=_I32, device=states.device)
        white = torch.empty(B, dtype=_I32, device=states.device)
    else:
        black, white = out
    code = _lib.lib().gg_batch_areas(_lib.dev_ptr(states, _U8, 'states'), _lib.dev_ptr(black, _I32, 'black'),
                                     _lib.dev_ptr(white, _I32, 'white'), B, N, _lib.stream_ptr(states.device))
    _lib.check(code, 'gg_batch_areas')
    return black, white


def _invalid_mask_dev(states, ko=None):
    B, C, N, _ = states.shape
    mask = torch.empty((B, N, N), dtype=_U8, device=states.device)
    code = _lib.lib().gg_batch_invalid_mask(_lib.dev_ptr(states, _U8, 'states'), _lib.dev_ptr(ko, _I32, 'ko'),
                                            _lib.dev_ptr(mask, _U8, 'mask'), B, N, _lib.stream_ptr(states.device))
    _lib.check(code, 'gg_batch_invalid_mask')
    return mask


# ------------------------------------------------------------------ gogame API

def init_state(size, device=None):
    """gym_go/gogame.py:22-25.  NumPy float64 zeros like the reference, or a device uint8 tensor."""
    if device is None:
        return np.zeros((govars.NUM_CHNLS, size, size))
    return torch.zeros((govars.NUM_CHNLS, size, size), dtype=_U8, device=device)


def batch_init_state(batch_size, board_size, device=None):
    """gym_go/gogame.py:28-31."""
    if device is None:
        return np.zeros((batch_size, govars.NUM_CHNLS, board_size, board_size))
    return torch.zeros((batch_size, govars.NUM_CHNLS, board_size, board_size), dtype=_U8, device=device)


def next_state(state, action1d, canonical=False):
    """gym_go/gogame.py:34-87.  Input is never mutated; illegal point -> AssertionError."""
    box = _Box(state)
    N = box.t.shape[-1]
    actions = torch.tensor([int(action1d)], dtype=_I32, device=box.t.device)
    out, status = _next_states_dev(box.t[None], actions, canonical)
    if int(status[0]) != 0:
        a = int(action1d)
        raise AssertionError(('Invalid move', (a // N, a % N)))
    return box.back(out[0])


def batch_next_states(batch_states, batch_action1d, canonical=False, check=True, out=None, status=None, workspace=None):
    """gym_go/gogame.py:90-150, with next_state's semantics for every game (also when the batch
    contains passes, where the reference mis-aligns games: gym_go/state_utils.py:187-193).
    check=True (default) synchronises to raise AssertionError like :117 if any move is illegal;
    check=False returns (next_states, status) without a host sync - illegal rows pass through.
    out / status (device tensors, optional): caller-owned result buffers - a per-ply loop that ping-pongs two
    state tensors then allocates nothing per call.
    workspace (gogame.next_states_workspace(B, N), optional): lets a loop that feeds each output back as the next
    input skip the from-scratch liberty analysis (gg_batch_next_states_ws): the classes of the last outputs are kept
    there and reused for every board whose stones match exactly; results are identical with and without it."""
    if (out is not None and isinstance(batch_states, torch.Tensor) and isinstance(batch_action1d, torch.Tensor)
            and batch_states.dtype == _U8 and batch_action1d.dtype == _I32 and not check):
        # hot loop: device tensors in the native dtypes, nothing to convert
        return _next_states_dev(batch_states, batch_action1d, canonical, out, status, workspace)
    box = _Box(batch_states)
    B = box.t.shape[0]
    actions = _actions_tensor(batch_action1d, B, box.t.device)
    out, status = _next_states_dev(box.t, actions, canonical, out, status, workspace)
    if not check:
        return box.back(out), status
    if B and bool((status != 0).any()):
        raise AssertionError('Invalid move in batch at games %s' % torch.nonzero(status).flatten()[:8].tolist())
    return box.back(out)


def invalid_moves(state):
    """gym_go/gogame.py:153-157: plane 3 flattened + [0] for pass; all zeros once the game ended."""
    box = _Box(state)
    t = box.t
    n = t.shape[-1] * t.shape[-2] + 1
    res = torch.zeros(n, dtype=_U8, device=t.device)
    if not game_ended(t):
        res[:-1] = t[govars.INVD_CHNL].reshape(-1)
    return box.back(res, np.float64 if box.numpy else None)


def valid_moves(state):
    """gym_go/gogame.py:160-161."""
    return 1 - invalid_moves(state)


def batch_invalid_moves(batch_state):
    """gym_go/gogame.py:164-168 (no game-over special case, like the reference)."""
    box = _Box(batch_state)
    t = box.t
    n = t.shape[0]
    flat = t[:, govars.INVD_CHNL].reshape(n, -1)
    res = torch.cat([flat, torch.zeros((n, 1), dtype=_U8, device=t.device)], dim=1)
    return box.back(res, np.float64 if box.numpy else None)


def batch_valid_moves(batch_state):
    """gym_go/gogame.py:171-172."""
    return 1 - batch_invalid_moves(batch_state)


def children(state, canonical=False, padded=True):
    """gym_go/gogame.py:175-186.  padded=True -> [N*N+1, 6, N, N] with all-zero slots for invalid
    actions; padded=False -> only the valid actions' successors, ascending action order."""
    box = _Box(state)
    if not padded:      # gym_go/gogame.py:179: only the successors of the valid actions - written by the device as such
        return box.back(_children_compact_dev(box.t[None], canonical)[0])
    return box.back(_children_dev(box.t[None], canonical)[0])


def batch_children(batch_states, canonical=False, padded=True, out=None, offsets=None):
    """BASELINE.json config 5 (no reference counterpart: == stack(children(s) for s in states)).
    padded=True -> [B, N*N+1, 6, N, N]; out (optional): a caller-owned uint8 device tensor of that shape to expand into
    (6.4 GB at config 5).
    padded=False -> (children [total, 6, N, N], offsets int32 [B+1]): the un-padded children of every state concatenated
    (== cat(children(s, padded=False) for s in states)); state b's are children[offsets[b]:offsets[b+1]].  On mid-game
    19x19 positions a third of the padded bytes.  out / offsets (optional): caller-owned buffers (out: at least `total`
    boards, e.g. B * (N*N+1); then nothing is read back to the host)."""
    box = _Box(batch_states)
    if not padded:
        kids, offs = _children_compact_dev(box.t, canonical, offsets, out)
        if box.numpy:
            offs = offs.cpu().numpy()
            return box.back(kids[:int(offs[-1])]), offs
        return kids, offs
    return box.back(_children_dev(box.t, canonical, out))


def action_size(state=None, board_size: int = None):
    """gym_go/gogame.py:189-197."""
    if state is not None:
        m, n = state.shape[1:]
    elif board_size is not None:
        m, n = board_size, board_size
    else:
        raise RuntimeError('No argument passed')
    return m * n + 1


def _plane_any(state, chnl):
    if isinstance(state, torch.Tensor):
        return bool((state[chnl] == 1).any())
    return bool(np.max(np.asarray(state)[chnl] == 1))


def prev_player_passed(state):
    """gym_go/gogame.py:200-201."""
    return _plane_any(state, govars.PASS_CHNL)


def batch_prev_player_passed(batch_state):
    """gym_go/gogame.py:204-205."""
    if isinstance(batch_state, torch.Tensor):
        return batch_state[:, govars.PASS_CHNL].amax(dim=(1, 2)) == 1
    return np.max(batch_state[:, govars.PASS_CHNL], axis=(1, 2)) == 1


def game_ended(state):
    """gym_go/gogame.py:208-214: 0/1."""
    if isinstance(state, torch.Tensor):
        return int(bool((state[govars.DONE_CHNL] == 1).all()))
    m, n = state.shape[1:]
    return int(np.count_nonzero(np.asarray(state)[govars.DONE_CHNL] == 1) == m * n)


def batch_game_ended(batch_state):
    """gym_go/gogame.py:217-222."""
    if isinstance(batch_state, torch.Tensor):
        return batch_state[:, govars.DONE_CHNL].amax(dim=(1, 2))
    return np.max(batch_state[:, govars.DONE_CHNL], axis=(1, 2))


def areas(state):
    """gym_go/gogame.py:275-300 (Tromp-Taylor) -> (black_area, white_area)."""
    box = _Box(state)
    b, w = _areas_dev(box.t[None])
    if box.numpy:
        return np.float64(int(b[0])), np.float64(int(w[0]))
    return b[0], w[0]


def batch_areas(batch_state, out=None):
    """gym_go/gogame.py:303-310 -> two [B] arrays.  out (optional): (black, white) int32 device tensors to write into."""
    box = _Box(batch_state)
    b, w = _areas_dev(box.t, out)
    if box.numpy:
        return b.cpu().numpy().astype(np.float64), w.cpu().numpy().astype(np.float64)
    return b, w


def winning(state, komi=0):
    """gym_go/gogame.py:225-230: sign(black_area - white_area - komi)."""
    black_area, white_area = areas(state)
    if isinstance(black_area, torch.Tensor):
        return torch.sign(black_area.to(torch.float64) - white_area.to(torch.float64) - komi)
    return np.sign(black_area - white_area - komi)


def batch_winning(state, komi=0):
    """gym_go/gogame.py:233-238."""
    b, w = batch_areas(state)
    if isinstance(b, torch.Tensor):
        return torch.sign(b.to(torch.float64) - w.to(torch.float64) - komi)
    return np.sign(b - w - komi)


def turn(state):
    """gym_go/gogame.py:241-246."""
    if isinstance(state, torch.Tensor):
        return int(state[govars.TURN_CHNL].max())
    return int(np.max(np.asarray(state)[govars.TURN_CHNL]))


def batch_turn(batch_state):
    """gym_go/gogame.py:249-250."""
    if isinstance(batch_state, torch.Tensor):
        return batch_state[:, govars.TURN_CHNL].amax(dim=(1, 2)).to(_I64)
    return np.max(batch_state[:, govars.TURN_CHNL], axis=(1, 2)).astype(int)


def liberties(state):
    """gym_go/gogame.py:253-264: per colour, empty points next to ANY stone of that colour (a dilation
    of the whole colour, not per group).  Host-side convenience, tensor ops only."""
    box = _Box(state)
    t = box.t.to(torch.bool)
    empty = ~(t[govars.BLACK] | t[govars.WHITE])
    res = []
    for stones in (t[govars.BLACK], t[govars.WHITE]):
        grown = torch.zeros_like(stones)
        grown[1:] |= stones[:-1]
        grown[:-1] |= stones[1:]
        grown[:, 1:] |= stones[:, :-1]
        grown[:, :-1] |= stones[:, 1:]
        res.append(grown & empty)
    if box.numpy:
        return res[0].cpu().numpy(), res[1].cpu().numpy()
    return res[0], res[1]


def num_liberties(state):
    """gym_go/gogame.py:267-272."""
    b, w = liberties(state)
    return int(b.sum()), int(w.sum())


def canonical_form(state):
    """gym_go/gogame.py:313-321: a copy; colours swapped and turn cleared when white is to move."""
    if isinstance(state, torch.Tensor):
        s = state.clone()
        if turn(s) == govars.WHITE:
            s[[govars.BLACK, govars.WHITE]] = state[[govars.WHITE, govars.BLACK]]
            s[govars.TURN_CHNL] = 1 - s[govars.TURN_CHNL]
        return s
    s = np.copy(state)
    if turn(s) == govars.WHITE:
        s[[govars.BLACK, govars.WHITE]] = s[[govars.WHITE, govars.BLACK]]
        s[govars.TURN_CHNL] = 1 - s[govars.TURN_CHNL]
    return s


def batch_canonical_form(batch_state):
    """gym_go/gogame.py:324-337 (no host loop: one masked swap)."""
    if isinstance(batch_state, torch.Tensor):
        s = batch_state.clone()
        white = batch_state[:, govars.TURN_CHNL].amax(dim=(1, 2)) == govars.WHITE
        sel = white[:, None, None]
        s[:, govars.BLACK] = torch.where(sel, batch_state[:, govars.WHITE], batch_state[:, govars.BLACK])
        s[:, govars.WHITE] = torch.where(sel, batch_state[:, govars.BLACK], batch_state[:, govars.WHITE])
        s[:, govars.TURN_CHNL] = torch.where(sel, torch.zeros_like(s[:, govars.TURN_CHNL]), s[:, govars.TURN_CHNL])
        return s
    s = np.copy(batch_state)
    white = batch_turn(s) == govars.WHITE
    s[white, govars.BLACK], s[white, govars.WHITE] = batch_state[white, govars.WHITE], batch_state[white, govars.BLACK]
    s[white, govars.TURN_CHNL] = 0
    return s


def _orient(image, i):
    flip = torch.flip if isinstance(image, torch.Tensor) else (lambda x, dims: np.flip(x, dims[0]))
    rot = (lambda x: torch.rot90(x, 1, (1, 2))) if isinstance(image, torch.Tensor) else (
        lambda x: np.rot90(x, axes=(1, 2)))
    x = image
    if (i >> 0) % 2:
        x = flip(x, (2,))
    if (i >> 1) % 2:
        x = flip(x, (1,))
    if (i >> 2) % 2:
        x = rot(x)
    return x


def random_symmetry(image):
    """gym_go/gogame.py:340-359: one of the 8 dihedral views of a [C, N, N] image."""
    return _orient(image, int(np.random.randint(0, 8)))


def all_symmetries(image):
    """gym_go/gogame.py:362-382: all 8, same order (h-flip bit 0, v-flip bit 1, rot90 bit 2)."""
    return [_orient(image, i) for i in range(8)]


def random_weighted_action(move_weights):
    """gym_go/gogame.py:385-392: L1-normalise, then draw."""
    w = np.asarray(move_weights.cpu() if isinstance(move_weights, torch.Tensor) else move_weights, dtype=np.float64)
    w = w / np.abs(w).sum()
    return np.random.choice(np.arange(len(w)), p=w)


def random_action(state):
    """gym_go/gogame.py:395-404: uniform over plane-3-clear points and pass."""
    inv = state[govars.INVD_CHNL].reshape(-1)
    inv = inv.cpu().numpy() if isinstance(inv, torch.Tensor) else np.asarray(inv)
    return random_weighted_action(1 - np.append(inv, 0))


def str(state):
    """gym_go/gogame.py:407-468: Unicode board + turn / game state / areas footer."""
    s = state.cpu().numpy() if isinstance(state, torch.Tensor) else np.asarray(state)
    size = s.shape[1]
    rows = ['\t' + ''.join('{}'.format(i).ljust(2, ' ') for i in range(size))]
    for i in range(size):
        line = '{}\t'.format(i)
        for j in range(size):
            last = j == size - 1
            if s[0, i, j] == 1 or s[1, i, j] == 1:
                line += '○' if s[0, i, j] == 1 else '●'
                if not last:
                    line += '═' if i in (0, size - 1) else '─'
            elif i == 0:
                line += '╔═' if j == 0 else ('╗' if last else '╤═')
            elif i == size - 1:
                line += '╚═' if j == 0 else ('╝' if last else '╧═')
            else:
                line += '╟─' if j == 0 else ('╢' if last else '┼─')
        rows.append(line)
    black_area, white_area = areas(state)
    phase = 'END' if game_ended(state) else ('PASSED' if prev_player_passed(state) else 'ONGOING')
    rows.append('\tTurn: {}, Game State (ONGOING|PASSED|END): {}'.format('BLACK' if turn(state) == 0 else 'WHITE', phase))
    rows.append('\tBlack Area: {}, White Area: {}'.format(int(black_area), int(white_area)))
    return '\n'.join(rows) + '\n'


# ------------------------------------------------------------------ device rollout helpers (build-side)

def rng_seed(batch_size, base_seed=20260927, first_game=0, device=None):
    """Per-game generator states for batch_rollout / batch_sample_actions (include/gymgo_amd.h)."""
    device = device or _device()
    # uint64 storage as int64 tensor (torch has no general uint64 ops; only the bits matter)
    rng = torch.empty(batch_size, dtype=_I64, device=device)
    code = _lib.lib().gg_rng_seed(_lib.dev_ptr(rng, _I64, 'rng'), int(base_seed) & (2 ** 64 - 1), int(first_game),
                                  batch_size, _lib.stream_ptr(device))
    _lib.check(code, 'gg_rng_seed')
    return rng


def batch_rollout(batch_states, rng, plies, auto_reset=True, last_actions=None, steps_done=None):
    """IN PLACE: `plies` uniform-random steps per game with the board resident on-chip (gg_batch_rollout)."""
    B, C, N, _ = batch_states.shape
    code = _lib.lib().gg_batch_rollout(
        _lib.dev_ptr(batch_states, _U8, 'states'), _lib.dev_ptr(rng, _I64, 'rng'),
        _lib.dev_ptr(last_actions, _I32, 'last_actions'), _lib.dev_ptr(steps_done, _I64, 'steps_done'),
        B, N, int(plies), int(bool(auto_reset)), _lib.stream_ptr(batch_states.device))
    _lib.check(code, 'gg_batch_rollout')
    return batch_states


REWARD_METHODS = {'real': 0, 'heuristic': 1}   # GG_REWARD_* of include/gymgo_amd.h (gym_go/envs/go_env.py:11-17)


def batch_env_step(batch_states, actions=None, rng=None, komi=0.0, reward_method='real', auto_reset=True, out=None):
    """IN PLACE GoEnv.step (gym_go/envs/go_env.py:49-76) of every game in ONE launch (gg_batch_env_step): auto-reset,
    the action (`actions`, or drawn uniformly with `rng` when None), legality, next_state, game_ended and
    GoEnv.reward (:128-149).  -> (rewards float32 [B], dones uint8 [B], status int32 [B], taken int32 [B]);
    `out` = such a 4-tuple to write into (no allocation: the call is then capturable in a hipGraph as is)."""
    B, C, N, _ = batch_states.shape
    dev = batch_states.device
    if actions is None and rng is None:
        raise ValueError('batch_env_step needs actions or an rng state to draw them with')
    if out is None:
        out = (torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B, dtype=_U8, device=dev),
               torch.empty(B, dtype=_I32, device=dev), torch.empty(B, dtype=_I32, device=dev))
    rewards, dones, status, taken = out
    code = _lib.lib().gg_batch_env_step(
        _lib.dev_ptr(batch_states, _U8, 'states'), _lib.dev_ptr(actions, _I32, 'actions'),
        _lib.dev_ptr(rng, _I64, 'rng'), _lib.dev_ptr(rewards, torch.float32, 'rewards'),
        _lib.dev_ptr(dones, _U8, 'dones'), _lib.dev_ptr(status, _I32, 'status'), _lib.dev_ptr(taken, _I32, 'taken'),
        B, N, float(komi), REWARD_METHODS[reward_method], int(bool(auto_reset)), _lib.stream_ptr(dev))
    _lib.check(code, 'gg_batch_env_step')
    return out


def batch_sample_actions(batch_states, rng):
    """actions[b] ~ Uniform{valid actions incl. pass} on the device (GoEnv.uniform_random_action,
    gym_go/envs/go_env.py:78-81, for every game at once)."""
    B, C, N, _ = batch_states.shape
    actions = torch.empty(B, dtype=_I32, device=batch_states.device)
    code = _lib.lib().gg_batch_sample_actions(
        _lib.dev_ptr(batch_states, _U8, 'states'), _lib.dev_ptr(rng, _I64, 'rng'),
        _lib.dev_ptr(actions, _I32, 'actions'), B, N, _lib.stream_ptr(batch_states.device))
    _lib.check(code, 'gg_batch_sample_actions')
    return actions


def batch_reset_finished(batch_states):
    """IN PLACE: every finished game (plane 5 set) becomes init_state (GoVecEnv auto-reset, gg_batch_reset_finished)."""
    B, C, N, _ = batch_states.shape
    code = _lib.lib().gg_batch_reset_finished(_lib.dev_ptr(batch_states, _U8, 'states'), B, N,
                                              _lib.stream_ptr(batch_states.device))
    _lib.check(code, 'gg_batch_reset_finished')
    return batch_states


def packed_words(board_size):
    """uint32 words per board of the bit-packed format (3 N + 1)."""
    return 3 * board_size + 1


def batch_pack(batch_states):
    """[B,6,N,N] uint8 device tensor -> [B, 3N+1] int32 device tensor (row masks of planes 0/1/3 + flag word);
    9.3x smaller at 19x19 - replay buffers, checkpoints, the wire (gg_batch_pack_states)."""
    B, C, N, _ = batch_states.shape
    packed = torch.empty((B, packed_words(N)), dtype=_I32, device=batch_states.device)
    code = _lib.lib().gg_batch_pack_states(_lib.dev_ptr(batch_states, _U8, 'states'), _lib.dev_ptr(packed, _I32, 'packed'),
                                           B, N, _lib.stream_ptr(batch_states.device))
    _lib.check(code, 'gg_batch_pack_states')
    return packed


def batch_unpack(packed, board_size):
    """Inverse of batch_pack (gg_batch_unpack_states)."""
    B = packed.shape[0]
    if packed.shape[1] != packed_words(board_size):
        raise ValueError('packed rows must have %d words for a %dx%d board' % (packed_words(board_size), board_size, board_size))
    states = torch.empty((B, govars.NUM_CHNLS, board_size, board_size), dtype=_U8, device=packed.device)
    code = _lib.lib().gg_batch_unpack_states(_lib.dev_ptr(packed, _I32, 'packed'), _lib.dev_ptr(states, _U8, 'states'),
                                             B, board_size, _lib.stream_ptr(packed.device))
    _lib.check(code, 'gg_batch_unpack_states')
    return states


# ---------------------------------------------------------------- the step path on packed boards
# Same operations as above on [B, 3N+1] int32 packed boards (batch_pack): 232 B per 19x19 board instead of 2 166 B and no
# byte <-> bit conversion inside the kernels - for search trees / replay buffers that unpack only what a network reads.

def _packed_size(packed):
    W = packed.shape[-1]
    if packed.dtype != _I32 or (W - 1) % 3 or not 2 <= (W - 1) // 3 <= 19:
        raise ValueError('packed boards are int32 [..., 3N+1] (got %s %s)' % (packed.dtype, tuple(packed.shape)))
    return (W - 1) // 3


def batch_next_states_packed(packed, batch_action1d, canonical=False, check=True):
    """gogame.batch_next_states (gym_go/gogame.py:90-150) on packed boards -> (packed_out, status int32 [B])."""
    N = _packed_size(packed)
    B = packed.shape[0]
    actions = _actions_tensor(batch_action1d, B, packed.device)
    out = torch.empty_like(packed)
    status = torch.empty(B, dtype=_I32, device=packed.device)
    code = _lib.lib().gg_batch_next_states_packed(
        _lib.dev_ptr(packed, _I32, 'packed'), _lib.dev_ptr(actions, _I32, 'actions'), _lib.dev_ptr(out, _I32, 'out'),
        _lib.dev_ptr(status, _I32, 'status'), B, N, int(bool(canonical)), _lib.stream_ptr(packed.device))
    _lib.check(code, 'gg_batch_next_states_packed')
    if check and bool((status != 0).any()):
        raise AssertionError('invalid move in batch (gym_go/gogame.py:117)')
    return out, status


def batch_rollout_packed(packed, rng, plies, auto_reset=True, last_actions=None, steps_done=None):
    """IN PLACE batch_rollout on packed boards (gg_batch_rollout_packed)."""
    N = _packed_size(packed)
    B = packed.shape[0]
    code = _lib.lib().gg_batch_rollout_packed(
        _lib.dev_ptr(packed, _I32, 'packed'), _lib.dev_ptr(rng, _I64, 'rng'), _lib.dev_ptr(last_actions, _I32, 'last_actions'),
        _lib.dev_ptr(steps_done, _I64, 'steps_done'), B, N, int(plies), int(bool(auto_reset)), _lib.stream_ptr(packed.device))
    _lib.check(code, 'gg_batch_rollout_packed')
    return packed


def batch_env_step_packed(packed, actions=None, rng=None, komi=0.0, reward_method='real', auto_reset=True, out=None):
    """IN PLACE batch_env_step on packed boards -> (rewards, dones, status, taken)."""
    N = _packed_size(packed)
    B = packed.shape[0]
    dev = packed.device
    if actions is None and rng is None:
        raise ValueError('batch_env_step_packed needs actions or an rng state to draw them with')
    if out is None:
        out = (torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B, dtype=_U8, device=dev),
               torch.empty(B, dtype=_I32, device=dev), torch.empty(B, dtype=_I32, device=dev))
    rewards, dones, status, taken = out
    code = _lib.lib().gg_batch_env_step_packed(
        _lib.dev_ptr(packed, _I32, 'packed'), _lib.dev_ptr(actions, _I32, 'actions'), _lib.dev_ptr(rng, _I64, 'rng'),
        _lib.dev_ptr(rewards, torch.float32, 'rewards'), _lib.dev_ptr(dones, _U8, 'dones'), _lib.dev_ptr(status, _I32, 'status'),
        _lib.dev_ptr(taken, _I32, 'taken'), B, N, float(komi), REWARD_METHODS[reward_method], int(bool(auto_reset)),
        _lib.stream_ptr(dev))
    _lib.check(code, 'gg_batch_env_step_packed')
    return out


def batch_children_packed(packed, canonical=False):
    """gogame.children (gym_go/gogame.py:175-186) of every packed parent -> int32 [B, N*N+1, 3N+1], invalid slots zero."""
    N = _packed_size(packed)
    B = packed.shape[0]
    kids = torch.empty((B, N * N + 1, packed_words(N)), dtype=_I32, device=packed.device)
    code = _lib.lib().gg_batch_children_packed(_lib.dev_ptr(packed, _I32, 'packed'), _lib.dev_ptr(kids, _I32, 'children'),
                                               B, N, int(bool(canonical)), _lib.stream_ptr(packed.device))
    _lib.check(code, 'gg_batch_children_packed')
    return kids


def batch_play_moves(batch_states, moves):
    """IN PLACE: state[b] = next_state(state[b], moves[b, t]) for t = 0 .. T-1 in one launch (gg_batch_play_moves; a loop
    of gym_go/gogame.py:34-87).  `batch_states`: uint8 [B,6,N,N] or packed int32 [B,3N+1]; moves: int [B, T].
    -> played int32 [B]: moves applied per game (a game stops at its first illegal move or when it has ended)."""
    packed = batch_states.dim() == 2
    N = _packed_size(batch_states) if packed else batch_states.shape[-1]
    B = batch_states.shape[0]
    moves = moves.to(device=batch_states.device, dtype=_I32).contiguous()
    if moves.dim() != 2 or moves.shape[0] != B:
        raise ValueError('moves must be [B, T]')
    played = torch.empty(B, dtype=_I32, device=batch_states.device)
    fn = _lib.lib().gg_batch_play_moves_packed if packed else _lib.lib().gg_batch_play_moves
    code = fn(_lib.dev_ptr(batch_states, _I32 if packed else _U8, 'states'), _lib.dev_ptr(moves, _I32, 'moves'),
              _lib.dev_ptr(played, _I32, 'played'), B, N, moves.shape[1], _lib.stream_ptr(batch_states.device))
    _lib.check(code, 'gg_batch_play_moves')
    return played


# ---------------------------------------------------------------- tracked boards: packed boards + their liberty classes
def tracked_words(board_size):
    """uint32 words per tracked board (5 N + 1): rows of black, white, invalid, multi_black, multi_white + flags."""
    return 5 * board_size + 1


def _tracked_size(tracked):
    W = tracked.shape[-1]
    if tracked.dtype != _I32 or (W - 1) % 5 or not 2 <= (W - 1) // 5 <= 19:
        raise ValueError('tracked boards are int32 [..., 5N+1] (got %s %s)' % (tracked.dtype, tuple(tracked.shape)))
    return (W - 1) // 5


def batch_track(batch_states):
    """uint8 [B,6,N,N] -> tracked int32 [B, 5N+1] (gg_batch_track_states): the packed board plus the stones of either
    colour whose group has >= 2 liberties.  Tracked boards step at the fused kernel's rate even one ply per launch."""
    B, C, N, _ = batch_states.shape
    tracked = torch.empty((B, tracked_words(N)), dtype=_I32, device=batch_states.device)
    code = _lib.lib().gg_batch_track_states(_lib.dev_ptr(batch_states, _U8, 'states'), _lib.dev_ptr(tracked, _I32, 'tracked'),
                                            B, N, _lib.stream_ptr(batch_states.device))
    _lib.check(code, 'gg_batch_track_states')
    return tracked


def batch_untrack(tracked, out=None):
    """tracked int32 [B, 5N+1] -> uint8 [B,6,N,N] (gg_batch_untrack_states); out: the tensor to write into (optional)."""
    N = _tracked_size(tracked)
    B = tracked.shape[0]
    states = out if out is not None else torch.empty((B, govars.NUM_CHNLS, N, N), dtype=_U8, device=tracked.device)
    if tuple(states.shape) != (B, govars.NUM_CHNLS, N, N):
        raise ValueError('out must be uint8 [B, 6, N, N]')
    code = _lib.lib().gg_batch_untrack_states(_lib.dev_ptr(tracked, _I32, 'tracked'), _lib.dev_ptr(states, _U8, 'states'),
                                              B, N, _lib.stream_ptr(tracked.device))
    _lib.check(code, 'gg_batch_untrack_states')
    return states


def batch_rollout_tracked(tracked, rng, plies, auto_reset=True, last_actions=None, steps_done=None):
    """IN PLACE batch_rollout on tracked boards (gg_batch_rollout_tracked)."""
    N = _tracked_size(tracked)
    B = tracked.shape[0]
    code = _lib.lib().gg_batch_rollout_tracked(
        _lib.dev_ptr(tracked, _I32, 'tracked'), _lib.dev_ptr(rng, _I64, 'rng'), _lib.dev_ptr(last_actions, _I32, 'last_actions'),
        _lib.dev_ptr(steps_done, _I64, 'steps_done'), B, N, int(plies), int(bool(auto_reset)), _lib.stream_ptr(tracked.device))
    _lib.check(code, 'gg_batch_rollout_tracked')
    return tracked


def batch_play_moves_tracked(tracked, moves, played=None):
    """IN PLACE batch_play_moves on tracked boards; moves [B, T] (T = 1: one GoEnv.step per game) -> played int32 [B]."""
    N = _tracked_size(tracked)
    B = tracked.shape[0]
    moves = moves.to(device=tracked.device, dtype=_I32).reshape(B, -1).contiguous()
    if played is None:
        played = torch.empty(B, dtype=_I32, device=tracked.device)
    code = _lib.lib().gg_batch_play_moves_tracked(_lib.dev_ptr(tracked, _I32, 'tracked'), _lib.dev_ptr(moves, _I32, 'moves'),
                                                  _lib.dev_ptr(played, _I32, 'played'), B, N, moves.shape[1],
                                                  _lib.stream_ptr(tracked.device))
    _lib.check(code, 'gg_batch_play_moves_tracked')
    return played


def batch_env_step_tracked(tracked, actions=None, rng=None, komi=0.0, reward_method='real', auto_reset=True, out=None,
                           states_out=None, steps_done=None, weights=None):
    """IN PLACE GoEnv.step (gym_go/envs/go_env.py:49-76) of every game on TRACKED boards in ONE launch
    (gg_batch_env_step_tracked): no per-ply analysis.  -> (rewards, dones, status, taken) like batch_env_step;
    states_out: a uint8 [B,6,N,N] device tensor that receives the byte-plane observation of every game (optional);
    steps_done: an int64 [B] device tensor, += 1 for every game whose step was played (optional).
    weights: float32 [B, N*N+1] policy weights - the move of every game is then DRAWN from them by the same launch
    (gg_batch_env_step_tracked_weighted: gogame.random_weighted_action, gym_go/gogame.py:385-392, masked by the game's
    invalid moves, with `rng`); `taken` receives the drawn moves, a game without a positive playable weight is refused."""
    N = _tracked_size(tracked)
    B = tracked.shape[0]
    dev = tracked.device
    if weights is not None:
        if actions is not None:
            raise ValueError('give actions OR weights, not both')
        if rng is None:
            raise ValueError('drawing from weights needs the rng state')
        weights, wcode = _weights_tensor(weights, B, N, dev)
    if actions is None and rng is None:
        raise ValueError('batch_env_step_tracked needs actions or an rng state to draw them with')
    if out is None:
        out = (torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B, dtype=_U8, device=dev),
               torch.empty(B, dtype=_I32, device=dev), torch.empty(B, dtype=_I32, device=dev))
    if states_out is not None and tuple(states_out.shape) != (B, govars.NUM_CHNLS, N, N):
        raise ValueError('states_out must be uint8 [B, 6, N, N]')
    rewards, dones, status, taken = out
    if weights is not None:
        code = _lib.lib().gg_batch_env_step_tracked_weighted(
            _lib.dev_ptr(tracked, _I32, 'tracked'), _lib.dev_ptr(weights, weights.dtype, 'weights'), wcode, _lib.dev_ptr(rng, _I64, 'rng'),
            _lib.dev_ptr(rewards, torch.float32, 'rewards'), _lib.dev_ptr(dones, _U8, 'dones'), _lib.dev_ptr(status, _I32, 'status'),
            _lib.dev_ptr(taken, _I32, 'taken'), _lib.dev_ptr(states_out, _U8, 'states_out'),
            _lib.dev_ptr(steps_done, _I64, 'steps_done'), B, N, float(komi),
            REWARD_METHODS[reward_method], int(bool(auto_reset)), _lib.stream_ptr(dev))
        _lib.check(code, 'gg_batch_env_step_tracked_weighted')
        return out
    code = _lib.lib().gg_batch_env_step_tracked(
        _lib.dev_ptr(tracked, _I32, 'tracked'), _lib.dev_ptr(actions, _I32, 'actions'), _lib.dev_ptr(rng, _I64, 'rng'),
        _lib.dev_ptr(rewards, torch.float32, 'rewards'), _lib.dev_ptr(dones, _U8, 'dones'), _lib.dev_ptr(status, _I32, 'status'),
        _lib.dev_ptr(taken, _I32, 'taken'), _lib.dev_ptr(states_out, _U8, 'states_out'),
        _lib.dev_ptr(steps_done, _I64, 'steps_done'), B, N, float(komi),
        REWARD_METHODS[reward_method], int(bool(auto_reset)), _lib.stream_ptr(dev))
    _lib.check(code, 'gg_batch_env_step_tracked')
    return out


# ---------------------------------------------------------------- policy-weighted sampling on the device
# gogame.random_weighted_action / random_action (gym_go/gogame.py:385-404) for every game of a batch: what a self-play loop
# with a policy network calls after the forward pass.  The draw is defined in integers (include/gymgo_amd.h,
# gg_batch_sample_weighted) so that device and CPU restatement agree bit for bit; P(a) = w[a] / sum(w) over the playable
# actions up to 22-bit fixed point relative to the largest weight.

WEIGHT_DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}   # GG_W_* of include/gymgo_amd.h


def _weights_tensor(weights, B, N, device):
    """-> (contiguous device tensor in its own dtype if that is float32 / bfloat16 / float16 - anything else becomes
    float32 - and the GG_W_* code).  The 16-bit forms are widened exactly by the kernels: half the bytes, the same draw."""
    if not isinstance(weights, torch.Tensor):
        weights = torch.as_tensor(np.asarray(weights, dtype=np.float32))
    if weights.dtype not in WEIGHT_DTYPES:
        weights = weights.to(torch.float32)
    w = weights.to(device=device).contiguous()
    if tuple(w.shape) != (B, N * N + 1):
        raise ValueError('weights must be [B, N*N+1] = [%d, %d] (got %s)' % (B, N * N + 1, tuple(w.shape)))
    return w, WEIGHT_DTYPES[w.dtype]


def _check_drawn(actions, check):
    if check and bool((actions < 0).any()):
        raise ValueError('no positive weight on a playable action for some game (np.random.choice raises here too)')
    return actions


def batch_sample_weighted(batch_states, weights, rng, check=False):
    """actions[b] ~ weights[b] / sum(weights[b]) over the actions plane 3 of batch_states[b] allows (the pass always;
    a finished game allows everything) - gogame.random_weighted_action (gym_go/gogame.py:385-392) per game, on the device,
    with the per-game generator `rng` (advanced once).  batch_states=None: nothing is masked (the reference "assumes all
    invalid moves have weight 0").  A game without a positive playable weight gets -1 (check=True: ValueError)."""
    if batch_states is not None:
        B, C, N, _ = batch_states.shape
        dev = batch_states.device
    else:
        if not isinstance(weights, torch.Tensor):
            raise ValueError('without states the weights must be a device tensor')
        B, dev = weights.shape[0], weights.device
        N = int(round((weights.shape[1] - 1) ** 0.5))
    w, wcode = _weights_tensor(weights, B, N, dev)
    actions = torch.empty(B, dtype=_I32, device=dev)
    code = _lib.lib().gg_batch_sample_weighted(
        _lib.dev_ptr(batch_states, _U8, 'states'), _lib.dev_ptr(w, w.dtype, 'weights'), wcode, _lib.dev_ptr(rng, _I64, 'rng'),
        _lib.dev_ptr(actions, _I32, 'actions'), B, N, _lib.stream_ptr(dev))
    _lib.check(code, 'gg_batch_sample_weighted')
    return _check_drawn(actions, check)


def batch_sample_weighted_rows(boards, board_size, weights, rng, check=False):
    """The same draw for packed (int32 [B, 3N+1]) or tracked (int32 [B, 5N+1]) boards."""
    B, W = boards.shape
    N = int(board_size)
    planes = (W - 1) // N if N > 0 else 0
    if boards.dtype != _I32 or planes * N + 1 != W or planes not in (3, 5):
        raise ValueError('boards must be packed [B, 3N+1] or tracked [B, 5N+1] int32 for N = %d (got %s)' % (N, tuple(boards.shape)))
    w, wcode = _weights_tensor(weights, B, N, boards.device)
    actions = torch.empty(B, dtype=_I32, device=boards.device)
    code = _lib.lib().gg_batch_sample_weighted_rows(
        _lib.dev_ptr(boards, _I32, 'boards'), planes, _lib.dev_ptr(w, w.dtype, 'weights'), wcode, _lib.dev_ptr(rng, _I64, 'rng'),
        _lib.dev_ptr(actions, _I32, 'actions'), B, N, _lib.stream_ptr(boards.device))
    _lib.check(code, 'gg_batch_sample_weighted_rows')
    return _check_drawn(actions, check)


def batch_random_action(batch_states, rng):
    """gogame.random_action (gym_go/gogame.py:395-404) per game: weights 1 - invalid_moves, i.e. uniform over the playable
    actions, through the weighted sampler (batch_sample_actions draws the same distribution with a cheaper kernel)."""
    B, C, N, _ = batch_states.shape
    ones = torch.ones((B, N * N + 1), dtype=torch.float32, device=batch_states.device)
    return batch_sample_weighted(batch_states, ones, rng)


# ---------------------------------------------------------------- batched symmetries on the device
# gogame.all_symmetries / random_symmetry (gym_go/gogame.py:340-382) for whole batches: one orientation per game or all
# eight, on byte planes (any channel count) and on packed / tracked boards.

def batch_symmetry(batch_images, orient=None, out=None):
    """uint8 [B, C, N, N] device tensor -> the view `orient[b]` (int32 [B], 0..7, composed as the reference does: bit 0
    flip the columns, bit 1 flip the rows, bit 2 rot90) of every image: [B, C, N, N]; orient=None: all eight views,
    [B, 8, C, N, N] in the order of all_symmetries (gg_batch_symmetry)."""
    B, C, N, N2 = batch_images.shape
    if N != N2:
        raise ValueError('images must be [B, C, N, N]')
    dev = batch_images.device
    shape = (B, C, N, N) if orient is not None else (B, 8, C, N, N)
    if out is None:
        out = torch.empty(shape, dtype=_U8, device=dev)
    elif tuple(out.shape) != shape:
        raise ValueError('out must be uint8 %s' % (shape,))
    if orient is not None:
        orient = _actions_tensor(orient, B, dev)
    code = _lib.lib().gg_batch_symmetry(_lib.dev_ptr(batch_images, _U8, 'images'), _lib.dev_ptr(orient, _I32, 'orient'),
                                        _lib.dev_ptr(out, _U8, 'out'), B, C, N, _lib.stream_ptr(dev))
    _lib.check(code, 'gg_batch_symmetry')
    return out


def batch_symmetry_rows(boards, board_size, orient=None):
    """The same on packed (int32 [B, 3N+1]) / tracked (int32 [B, 5N+1]) boards: -> [B, W], or [B, 8, W] for all eight
    views (gg_batch_symmetry_rows).  A transformed tracked board is a valid tracked board."""
    B, W = boards.shape
    N = int(board_size)
    planes = (W - 1) // N if N > 0 else 0
    if boards.dtype != _I32 or planes * N + 1 != W or planes not in (3, 5):
        raise ValueError('boards must be packed [B, 3N+1] or tracked [B, 5N+1] int32 for N = %d (got %s)' % (N, tuple(boards.shape)))
    out = torch.empty((B, W) if orient is not None else (B, 8, W), dtype=_I32, device=boards.device)
    if orient is not None:
        orient = _actions_tensor(orient, B, boards.device)
    code = _lib.lib().gg_batch_symmetry_rows(_lib.dev_ptr(boards, _I32, 'boards'), planes, _lib.dev_ptr(orient, _I32, 'orient'),
                                             _lib.dev_ptr(out, _I32, 'out'), B, N, _lib.stream_ptr(boards.device))
    _lib.check(code, 'gg_batch_symmetry_rows')
    return out


def batch_random_symmetry(batch_images, generator=None):
    """gogame.random_symmetry (gym_go/gogame.py:340-359) per game: -> (views [B, C, N, N], orient int32 [B]); the
    orientations come from torch's device generator (pass `generator` for reproducibility)."""
    B = batch_images.shape[0]
    orient = torch.randint(0, 8, (B,), dtype=_I32, device=batch_images.device, generator=generator)
    return batch_symmetry(batch_images, orient), orient


def symmetry_actions(actions, orient, board_size):
    """Where a move lands under the views above: action a of the ORIGINAL board -> the action that marks the same point
    on the view `orient` (the pass stays the pass).  int tensors / arrays [B] -> int32 tensor [B] on actions' device."""
    N = int(board_size)
    a = actions if isinstance(actions, torch.Tensor) else torch.as_tensor(np.asarray(actions))
    o = orient if isinstance(orient, torch.Tensor) else torch.as_tensor(np.asarray(orient))
    a = a.to(torch.int64)
    o = o.to(device=a.device, dtype=torch.int64)
    sr, sc = torch.div(a, N, rounding_mode='floor'), a % N
    r1 = torch.where((o & 2) != 0, N - 1 - sr, sr)
    c1 = torch.where((o & 1) != 0, N - 1 - sc, sc)
    r = torch.where((o & 4) != 0, N - 1 - c1, r1)
    c = torch.where((o & 4) != 0, r1, c1)
    return torch.where(a >= N * N, a, r * N + c).to(_I32)
