"""GoEnv with the reference's interface (gym_go/envs/go_env.py:19-152) on the HIP backend.

One game, NumPy float64 states in and out like the reference (config 1 "plumbing").  The game's record - action, state,
areas, status, done - lives in ONE block of pinned host memory that the device maps (hipHostMalloc through torch's
pinned allocator): step() writes the action into it, enqueues ONE launch (gg_batch_env_step_scored: legality, transition,
game-over flag and the Tromp-Taylor score the reward needs; the kernel reads and writes the record in place over the host
link) and waits for the stream - no host<->device copy in either direction, nothing computed on the host.  (Round 5: one
H2D copy, two launches and one D2H copy per step.)  Rendering (pyglet UI, gym_go/envs/go_env.py:160-243) is out of scope;
render('terminal') prints.
"""
from enum import Enum

import numpy as np

from gymgo_amd import gogame, govars
from gymgo_amd.envs import spaces


class RewardMethod(Enum):
    """REAL: 0 while playing, then sign(black area - white area - komi) (a tie gives 0, as the code at
    gym_go/gogame.py:230 does).  HEURISTIC: area difference - komi while playing, +-size**2 at the
    end (a tie gives -size**2, gym_go/envs/go_env.py:145-146)."""
    REAL = 'real'
    HEURISTIC = 'heuristic'


class GoEnv(spaces.Env):
    """gym.Env subclass when gym / gymnasium is importable (gym_go/envs/go_env.py:19), the same attribute surface
    through gymgo_amd.envs.spaces otherwise."""
    metadata = {'render.modes': ['terminal']}
    govars = govars
    gogame = gogame

    def __init__(self, size, komi=0, reward_method='real'):
        self.size = size
        self.komi = komi
        self.state_ = gogame.init_state(size)
        self.reward_method = RewardMethod(reward_method)
        # gym_go/envs/go_env.py:35-37
        self.observation_space = spaces.Box(np.float32(0), np.float32(govars.NUM_CHNLS),
                                            shape=(govars.NUM_CHNLS, size, size))
        self.action_space = spaces.Discrete(gogame.action_size(self.state_))
        self.done = False
        # `state_` is a public attribute in the reference, and callers both REPLACE it (`env.state_ = x`) and EDIT it in
        # place (`env.state_[0, 1, 1] = 1`); the reference recomputes from it on every call.  The device copy and the cached
        # score therefore remember the exact bytes they belong to (a uint8 snapshot, 6 N^2 bytes) and are refreshed
        # whenever `state_` no longer matches it.
        self._areas, self._areas_of = (0.0, 0.0), self._snapshot()
        self._dev = None          # device record, allocated on the first step
        self._dev_of = None       # snapshot of the host state the device copy mirrors

    def _snapshot(self):
        return np.ascontiguousarray(self.state_).astype(np.uint8)

    def _same(self, snap):
        return snap is not None and snap.shape == np.shape(self.state_) and np.array_equal(snap, self.state_)

    # ---- the record the kernel works on, in pinned (device-mapped) host memory:
    #      [64 B pad | state 6 N^2 | pad to 16 | black i32 | white i32 | status i32 | done u8 ... | action i32 | 64 B pad]
    # (the kernels read whole aligned 16-byte vectors around a board: the pads keep those reads inside the block)
    def _record(self):
        import torch
        if self._dev is None:
            from gymgo_amd import _lib
            dev = gogame._device()
            if dev.type != 'cuda' or not torch.cuda.is_available():
                raise _lib.GymGoNativeError('GoEnv needs a ROCm device (gymgo_amd has no CPU path)')
            n2 = govars.NUM_CHNLS * self.size * self.size
            off = 64 + ((n2 + 15) & ~15)
            with torch.cuda.device(dev):
                buf = torch.zeros(off + 32 + 64, dtype=torch.uint8).pin_memory()
            host = buf.numpy()
            base = buf.data_ptr()       # pinned host memory is mapped into the device's address space at the same address
            self._dev = {
                'buf': buf, 'host': host, 'device': dev,
                'state': host[64:64 + n2].reshape(govars.NUM_CHNLS, self.size, self.size),
                'words': host[off:off + 12].view(np.int32),       # black, white, status
                'done': host[off + 12:off + 13], 'action': host[off + 16:off + 20].view(np.int32),
                'p_state': base + 64, 'p_areas': base + off, 'p_status': base + off + 8, 'p_done': base + off + 12,
                'p_action': base + off + 16,
            }
        if not self._same(self._dev_of):      # replaced or edited in place by the caller: hand the kernel the new bytes
            snap = self._snapshot()
            self._dev['state'][...] = snap
            self._dev_of = snap
        return self._dev

    def reset(self):
        self.state_ = gogame.init_state(self.size)
        self.done = False
        self._areas, self._areas_of = (0.0, 0.0), self._snapshot()
        return np.copy(self.state_)

    def step(self, action):
        """-> (state, reward, done, info).  int | (r, c) tuple / list / ndarray | None (= pass)."""
        assert not self.done
        if isinstance(action, (tuple, list, np.ndarray)):
            assert 0 <= action[0] < self.size
            assert 0 <= action[1] < self.size
            action = self.size * action[0] + action[1]
        elif action is None:
            action = self.size ** 2
        from gymgo_amd import _lib
        rec = self._record()
        n = self.size
        rec['action'][0] = int(action)
        stream = _lib.stream_ptr(rec['device'])
        _lib.check(_lib.lib().gg_batch_env_step_scored(rec['p_state'], rec['p_action'], None, None, rec['p_done'], rec['p_status'],
                                                       None, rec['p_areas'], 1, n, 0.0, 0, 0, stream), 'gg_batch_env_step_scored')
        _lib.check(_lib.hip_runtime().hipStreamSynchronize(stream), 'hipStreamSynchronize')   # the kernel's writes to the record are visible from here on
        black, white, status = (int(x) for x in rec['words'])
        if status != 0:                           # gym_go/gogame.py:59: the position is unchanged
            a = int(action)
            raise AssertionError(('Invalid move', (a // n, a % n)))
        snap = rec['state'].copy()
        self.state_ = snap.astype(np.float64)
        self._dev_of = snap
        self._areas, self._areas_of = (float(black), float(white)), snap
        self.done = int(rec['done'][0])
        return np.copy(self.state_), self.reward(), self.done, self.info()

    def game_ended(self):
        return self.done

    def turn(self):
        return gogame.turn(self.state_)

    def prev_player_passed(self):
        return gogame.prev_player_passed(self.state_)

    def valid_moves(self):
        return 1 - self.invalid_moves()

    def uniform_random_action(self):
        return np.random.choice(np.argwhere(self.valid_moves()).flatten())

    def invalid_moves(self):
        """gogame.invalid_moves (gym_go/gogame.py:153-157) read off the host copy of the state: plane 3 + [0] for the
        pass, all zeros once the game has ended."""
        if self.done:
            return np.zeros(self.size ** 2 + 1)
        return np.append(self.state_[govars.INVD_CHNL].flatten(), 0)

    def info(self):
        return {
            'turn': int(self.state_[govars.TURN_CHNL, 0, 0]),
            'invalid_moves': self.invalid_moves(),
            'prev_player_passed': bool(self.state_[govars.PASS_CHNL, 0, 0] == 1),
        }

    def state(self):
        return np.copy(self.state_)

    def canonical_state(self):
        return gogame.canonical_form(self.state_)

    def children(self, canonical=False, padded=True):
        return gogame.children(self.state_, canonical, padded)

    def _score(self):
        if not self._same(self._areas_of):     # state replaced or edited by the caller: score it afresh
            self._areas, self._areas_of = tuple(float(x) for x in gogame.areas(self.state_)), self._snapshot()
        return self._areas

    def winning(self):
        """Who leads from black's perspective, whether or not the game is over (gym_go/gogame.py:225-230)."""
        black_area, white_area = self._score()
        return np.sign(black_area - white_area - self.komi)

    def winner(self):
        return self.winning() if self.game_ended() else 0

    def reward(self):
        if self.reward_method == RewardMethod.REAL:
            return self.winner()
        if self.reward_method == RewardMethod.HEURISTIC:
            black_area, white_area = self._score()
            margin = black_area - white_area - self.komi
            if self.game_ended():
                return (1 if margin > 0 else -1) * self.size ** 2
            return margin
        raise Exception('Unknown Reward Method')

    def __str__(self):
        return gogame.str(self.state_)

    def close(self):
        pass

    def render(self, mode='terminal'):
        if mode != 'terminal':
            raise NotImplementedError("only mode='terminal' (the pyglet UI is out of scope)")
        print(self.__str__())
