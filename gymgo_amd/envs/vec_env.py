"""GoVecEnv - B independent games as ONE uint8 [B, 6, N, N] device tensor (no reference counterpart:
the reference has one game per GoEnv; this is the batched form its gogame.batch_* functions imply).

One process drives one GPU; to use several GPUs run one process per GPU with `shard(rank, world)`
slices of the game range - games never interact, so there is no collective on the data path.
"""
import torch

from gymgo_amd import gogame, govars


def shard(total_games, rank, world_size):
    """Contiguous equal split of the game index range [0, total_games) -> (first_game, count)."""
    base, rem = divmod(total_games, world_size)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


class GoVecEnv:
    """packed=True keeps the boards bit-packed on the device ([B, 3N+1] int32, 232 B per 19x19 board) and steps them
    with the gg_batch_*_packed kernels (1.7x the per-ply rate); `states` then unpacks a fresh uint8 [B,6,N,N] view
    on demand (e.g. as network input), `packed_states` is the resident tensor."""

    def __init__(self, batch_size, size, komi=0, reward_method='real', device=None, seed=20260927, first_game=0,
                 auto_reset=True, packed=False):
        self.batch_size, self.size, self.komi = batch_size, size, komi
        self.reward_method = reward_method
        self.device = torch.device(device) if device is not None else gogame._device()
        self.auto_reset = auto_reset
        self.packed = bool(packed)
        if self.packed:
            self.packed_states = torch.zeros((batch_size, gogame.packed_words(size)), dtype=torch.int32, device=self.device)
        else:
            self._states = gogame.batch_init_state(batch_size, size, device=self.device)
        self.rng = gogame.rng_seed(batch_size, seed, first_game, self.device)
        self.steps_done = torch.zeros(batch_size, dtype=torch.int64, device=self.device)
        self.last_actions = torch.full((batch_size,), -1, dtype=torch.int32, device=self.device)
        # step() outputs live in fixed buffers (rewards, dones, status; the action taken goes to last_actions): no
        # allocation per step, and the step can be captured in a hipGraph.  They are overwritten by the next step.
        self._step_out = (torch.empty(batch_size, dtype=torch.float32, device=self.device),
                          torch.empty(batch_size, dtype=torch.uint8, device=self.device),
                          torch.empty(batch_size, dtype=torch.int32, device=self.device), self.last_actions)

    @property
    def states(self):
        return gogame.batch_unpack(self.packed_states, self.size) if self.packed else self._states

    @states.setter
    def states(self, value):
        if self.packed:
            self.packed_states = gogame.batch_pack(value)
        else:
            self._states = value

    def reset(self, mask=None):
        store = self.packed_states if self.packed else self._states
        if mask is None:
            store.zero_()
        else:
            store[mask] = 0
        return self.states

    def valid_moves(self):
        return gogame.batch_valid_moves(self.states)

    def sample_actions(self):
        """Uniform over valid actions incl. pass, per game, on the device."""
        return gogame.batch_sample_actions(self.states, self.rng)

    def step(self, actions=None, check=False):
        """One GoEnv.step per game in ONE launch -> (states, rewards, dones, status); `states` is the resident tensor
        (the packed one when packed=True).  actions=None draws a
        uniform-random valid action per game on the device (it is left in self.last_actions).  Finished games are
        reset first when auto_reset; rewards are float32, black's perspective (gym_go/envs/go_env.py:128-149).
        The returned rewards / dones / status are views of fixed buffers, overwritten by the next step()."""
        if actions is not None:
            actions = actions.to(device=self.device, dtype=torch.int32).contiguous()
        fn, store = ((gogame.batch_env_step_packed, self.packed_states) if self.packed
                     else (gogame.batch_env_step, self._states))
        rewards, dones, status, taken = fn(store, actions, self.rng, self.komi, self.reward_method, self.auto_reset,
                                           out=self._step_out)
        if check and bool((status != 0).any()):
            raise AssertionError('illegal move in batch')
        self.steps_done += (status == 0)
        return (self.packed_states if self.packed else self._states), rewards, dones, status

    def step_unfused(self, actions, check=False):
        """The same step as separate launches (reset, next_states, areas + torch reward arithmetic); float64 rewards."""
        if self.packed:
            raise NotImplementedError('step_unfused works on byte-plane states (packed=False)')
        if self.auto_reset:
            gogame.batch_reset_finished(self.states)
        actions = actions.to(device=self.device, dtype=torch.int32)
        self.states, status = gogame.batch_next_states(self.states, actions, check=False)
        if check and bool((status != 0).any()):
            raise AssertionError('illegal move in batch')
        self.last_actions.copy_(actions)
        self.steps_done += (status == 0).to(torch.int64)
        dones = self.states[:, govars.DONE_CHNL, 0, 0]   # planes 2/4/5 are uniform: one byte per game
        return self.states, self.rewards(dones), dones, status

    def rollout(self, plies):
        """`plies` uniform-random steps per game, fused on the device (board stays on-chip)."""
        if self.packed:
            gogame.batch_rollout_packed(self.packed_states, self.rng, plies, self.auto_reset, self.last_actions, self.steps_done)
            return self.packed_states
        gogame.batch_rollout(self._states, self.rng, plies, self.auto_reset, self.last_actions, self.steps_done)
        return self._states

    def rewards(self, dones=None):
        """GoEnv.reward for every game (gym_go/envs/go_env.py:128-149), black's perspective."""
        black, white = gogame.batch_areas(self.states)
        margin = black.to(torch.float64) - white.to(torch.float64) - self.komi
        if dones is None:
            dones = self.states[:, govars.DONE_CHNL, 0, 0]
        over = dones.bool()
        if self.reward_method == 'real':
            return torch.where(over, torch.sign(margin), torch.zeros_like(margin))
        final = torch.where(margin > 0, 1.0, -1.0).to(torch.float64) * self.size ** 2
        return torch.where(over, final, margin)

    def turns(self):
        return self.states[:, govars.TURN_CHNL, 0, 0]
