"""The two UNBATCHED stack tasks of the reference's task map (num_envs = 0; /root/reference/gym_genesis/env.py:110-117):
``FrankaCubeStackOne`` (/root/reference/gym_genesis/tasks/franka/cube_stack_one.py:28-198) and ``CubeStackOne``
(/root/reference/gym_genesis/tasks/so101/cube_stack.py:25-207).  They build the same scenes without an env axis and
return observations without a batch dimension ((9,)/(6,) and (14,)), a scalar reward, and draw their spawn positions as
scalars: x1, y1, x2, y2 (SO-101: re-drawn until 6 cm apart), then xd, yd per distractor -- the same values, in the same
order, as the batched classes draw with B = 1.  So each is the batched class run with one env and its leading axis
squeezed; ``action`` is a (n,) vector.  ``num_envs`` reads 0 as in the reference.
"""
from __future__ import annotations

import numpy as np
import torch

from .franka.cube_stack_kitchen_batch import FrankaCubeStackKitchenBatch
from .so101.cube_stack_batch import CubeStackBatch


class _One:
    def __init__(self, enable_pixels, observation_height, observation_width, num_envs, env_spacing, camera_capture_mode,
                 strip_environment_state, shard=None):
        super().__init__(enable_pixels, observation_height, observation_width, 1, env_spacing, camera_capture_mode,
                         strip_environment_state)
        self.batch_envs = 1
        self.unbatched = True

    @staticmethod
    def _squeeze(obs):
        out = {}
        for k, v in obs.items():
            out[k] = {n: t[0] for n, t in v.items()} if isinstance(v, dict) else (v[0] if v.dim() > 3 or k != "pixels" else v)
        return out

    # reset() is inherited: it ends in self.get_obs(), which squeezes

    def get_obs(self):
        return self._squeeze(super().get_obs())

    def step(self, action):
        a = torch.as_tensor(np.asarray(action) if not isinstance(action, torch.Tensor) else action).reshape(1, -1)
        _, reward, _, obs = super().step(a)
        return None, reward[0], None, self._squeeze(obs)

    def compute_reward(self):
        return super().compute_reward()[0]


class FrankaCubeStackOne(_One, FrankaCubeStackKitchenBatch):
    pass


class CubeStackOne(_One, CubeStackBatch):
    pass
