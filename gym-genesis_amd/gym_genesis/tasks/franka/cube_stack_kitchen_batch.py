"""Batched Franka cube-stack task in the kitchen scene on the MI355X backend (gym_genesis/CubeStack-v0, robot="franka").

Behavioural contract restated from /root/reference/gym_genesis/tasks/franka/cube_stack_kitchen_batch.py:27-224 and the
scene builder /root/reference/gym_genesis/tasks/utils.py:239-426:
  * scene: island slab (top z = 0.70003), Panda x0.6 at (-0.5, 0, 0.7), five 4 cm cubes on the slab          (utils.py:354-423)
  * reset(): z = island_top + 0.021, all cube quats (0,0,0,1) wxyz; draws from the task RandomState IN THIS ORDER:
    x1 ~ U(-0.3,-0.1) (B), y1 ~ U(-0.15,0.15) (B), x2, y2 likewise, then per distractor xd ~ U(-0.35,0) (B),
    yd ~ U(-0.2,0.2) (B); arm at the home pose, zero velocity, PD targets = home; kp/kv/force ranges as listed;
    ONE physics step                                                                                           (:64-115)
  * step(a): targets <- a[:, :7], a[:, 7:], one physics step, returns (None, reward, None, obs)                (:131-137)
  * reward = float32(|cube1.xy - cube2.xy| < 0.05 and cube1.z - cube2.z > 0.03)                                (:139-147)
  * obs: agent_pos (B,9) = [eef pos3, eef quat4, finger q2]; environment_state (B,14) = [cube1 pos3, cube1 quat4,
    eef - cube1 3, |eef - cube1| 1, cube2 pos3]                                                                (:149-167)
Differences by design: control + step + reward + observations are one launch of the wave-per-env kernel; `reward` is
a torch tensor on the device (the reference returns NumPy, which its own GenesisEnv.step cannot consume: SURVEY C-1).
"""
from __future__ import annotations

import numpy as np
import torch

from ...backend import models
from ..stack_common import StackTaskBase

AGENT_DIM = len(models.FRANKA_JOINTS)
ENV_DIM = 14


class FrankaCubeStackKitchenBatch(StackTaskBase):
    AGENT_DIM = AGENT_DIM
    ENV_DIM = ENV_DIM
    ROBOT_ROOT = "link0"
    JOINTS = models.FRANKA_JOINTS
    EEF_LINK = "hand"

    # cameras as created (utils.py:312-336) and as posed per env in get_obs() (cube_stack_kitchen_batch.py:175-182)
    CAM_TOP = ((0.0, 0.0, 1.5), (0.0, 0.0, 0.0), 40.0)
    CAM_SIDE = ((1.0, 0.0, 0.5), (0.0, 0.0, 0.5), 40.0)
    CAM_WRIST = ((0.4, 0.0, 0.7), (0.0, 0.0, 1.0), 90.0)
    PER_ENV_TOP = ((0.0, 0.0, 2.0), (0.0, 0.0, 0.5))
    PER_ENV_SIDE = ((1.5, 0.0, 0.8), (0.0, 0.0, 0.5))

    def _wrist_camera(self):
        # "wrist view (approximation, fixed offset)": at the hand link, looking along +x (:185-192)
        pos = self.eef.get_pos()
        look = pos + torch.tensor([0.1, 0.0, 0.0], device=pos.device)
        return pos, look, None, False

    def _scene_builder(self):
        return models.franka_cube_stack_scene()

    def _set_robot(self, view):
        self.franka = view
        self.motors_dof = np.arange(7)
        self.fingers_dof = np.arange(7, 9)

    def _home_qpos(self):
        return models.FRANKA_HOME  # cube_stack_kitchen_batch.py:94

    def sample_spawn(self) -> np.ndarray:
        Bg, r = self.global_num_envs, self._random
        z = np.full(Bg, self.island_top_z + 0.02 + 0.001)
        cols = []
        for _ in range(2):  # cube_1 then cube_2 (:71-82)
            x = r.uniform(-0.3, -0.1, size=(Bg,))
            y = r.uniform(-0.15, 0.15, size=(Bg,))
            cols.append(np.stack([x, y, z], axis=1))
        for _ in range(3):  # distractors (:85-91)
            x = r.uniform(-0.35, 0.0, size=(Bg,))
            y = r.uniform(-0.2, 0.2, size=(Bg,))
            cols.append(np.stack([x, y, z], axis=1))
        return np.stack(cols, axis=1).astype(np.float32)
