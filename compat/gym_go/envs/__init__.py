"""gym_go.envs (gym_go/envs/__init__.py): GoEnv of the MI355X backend under the reference's module path."""
import sys

from gymgo_amd.envs import GoEnv, GoVecEnv, GoVecEnvParts, RewardMethod, make  # noqa: F401
from gymgo_amd.envs import go_env

sys.modules.setdefault(__name__ + '.go_env', go_env)   # `from gym_go.envs.go_env import GoEnv, RewardMethod`
