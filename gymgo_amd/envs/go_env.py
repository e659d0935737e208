"""GoEnv with the reference's interface (gym_go/envs/go_env.py:19-152) on the HIP backend.

One game, NumPy float64 states in and out like the reference (config 1 "plumbing"); every
transition, children fan-out and area score goes through the device kernels via gymgo_amd.gogame.
Rendering (pyglet UI, gym_go/envs/go_env.py:160-243) is out of scope; render('terminal') prints.
"""
from enum import Enum

import numpy as np

from gymgo_amd import gogame, govars


class RewardMethod(Enum):
    """REAL: 0 while playing, then sign(black area - white area - komi) (a tie gives 0, as the code at
    gym_go/gogame.py:230 does).  HEURISTIC: area difference - komi while playing, +-size**2 at the
    end (a tie gives -size**2, gym_go/envs/go_env.py:145-146)."""
    REAL = 'real'
    HEURISTIC = 'heuristic'


class GoEnv:
    metadata = {'render.modes': ['terminal']}
    govars = govars
    gogame = gogame

    def __init__(self, size, komi=0, reward_method='real'):
        self.size = size
        self.komi = komi
        self.state_ = gogame.init_state(size)
        self.reward_method = RewardMethod(reward_method)
        self.observation_shape = (govars.NUM_CHNLS, size, size)
        self.action_n = gogame.action_size(self.state_)
        self.done = False

    def reset(self):
        self.state_ = gogame.init_state(self.size)
        self.done = False
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
        self.state_ = gogame.next_state(self.state_, action, canonical=False)
        self.done = gogame.game_ended(self.state_)
        return np.copy(self.state_), self.reward(), self.done, self.info()

    def game_ended(self):
        return self.done

    def turn(self):
        return gogame.turn(self.state_)

    def prev_player_passed(self):
        return gogame.prev_player_passed(self.state_)

    def valid_moves(self):
        return gogame.valid_moves(self.state_)

    def uniform_random_action(self):
        return np.random.choice(np.argwhere(self.valid_moves()).flatten())

    def info(self):
        return {
            'turn': gogame.turn(self.state_),
            'invalid_moves': gogame.invalid_moves(self.state_),
            'prev_player_passed': gogame.prev_player_passed(self.state_),
        }

    def state(self):
        return np.copy(self.state_)

    def canonical_state(self):
        return gogame.canonical_form(self.state_)

    def children(self, canonical=False, padded=True):
        return gogame.children(self.state_, canonical, padded)

    def winning(self):
        """Who leads from black's perspective, whether or not the game is over."""
        return gogame.winning(self.state_, self.komi)

    def winner(self):
        return self.winning() if self.game_ended() else 0

    def reward(self):
        if self.reward_method == RewardMethod.REAL:
            return self.winner()
        if self.reward_method == RewardMethod.HEURISTIC:
            black_area, white_area = gogame.areas(self.state_)
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
