"""GoEnv with the reference's interface (gym_go/envs/go_env.py:19-152) on the HIP backend.

One game, NumPy float64 states in and out like the reference (config 1 "plumbing").  The board lives on the device;
one step() is one host->device copy (the action), two launches (gg_batch_env_step: legality, transition, game-over
flag; gg_batch_areas: the Tromp-Taylor score the reward needs) and ONE device->host copy of a small record (state,
areas, done, status) from which the reward, `info()` and the returned state are read - nothing is computed on the host
and nothing is copied twice.  Rendering (pyglet UI, gym_go/envs/go_env.py:160-243) is out of scope; render('terminal') prints.
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

    # ---- the resident device record: [state 6 N^2 | pad | black i32 | white i32 | status i32 | done u8 ...]
    def _record(self):
        import torch
        if self._dev is None:
            n2 = govars.NUM_CHNLS * self.size * self.size
            off = (n2 + 15) & ~15
            buf = torch.zeros(off + 16, dtype=torch.uint8, device=gogame._device())
            self._dev = {
                'buf': buf, 'off': off,
                'states': buf[:n2].view(1, govars.NUM_CHNLS, self.size, self.size),
                'black': buf[off:off + 4].view(torch.int32), 'white': buf[off + 4:off + 8].view(torch.int32),
                'status': buf[off + 8:off + 12].view(torch.int32), 'done': buf[off + 12:off + 13],
            }
        if not self._same(self._dev_of):      # replaced or edited in place by the caller: upload it
            snap = self._snapshot()
            self._dev['states'].copy_(torch.from_numpy(snap).view(1, govars.NUM_CHNLS, self.size, self.size))
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
        import torch
        from gymgo_amd import _lib
        rec = self._record()
        n = self.size
        act = torch.tensor([int(action)], dtype=torch.int32, device=rec['buf'].device)
        gogame.batch_env_step(rec['states'], act, None, 0.0, 'real', False,
                              out=(None, rec['done'], rec['status'], None))
        L = _lib.lib()
        _lib.check(L.gg_batch_areas(_lib.dev_ptr(rec['states'], torch.uint8, 'states'),
                                    _lib.dev_ptr(rec['black'], torch.int32, 'black'),
                                    _lib.dev_ptr(rec['white'], torch.int32, 'white'), 1, n,
                                    _lib.stream_ptr(rec['buf'].device)), 'gg_batch_areas')
        host = rec['buf'].cpu().numpy()           # the one device->host copy of the step
        off = rec['off']
        black, white, status = (int(x) for x in host[off:off + 12].view(np.int32))
        if status != 0:                           # gym_go/gogame.py:59: the position is unchanged
            a = int(action)
            raise AssertionError(('Invalid move', (a // n, a % n)))
        snap = host[:off][:govars.NUM_CHNLS * n * n].reshape(govars.NUM_CHNLS, n, n).copy()
        self.state_ = snap.astype(np.float64)
        self._dev_of = snap
        self._areas, self._areas_of = (float(black), float(white)), snap
        self.done = int(host[off + 12])
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
