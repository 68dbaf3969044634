"""Batched SO-101 cube-stack task on the MI355X backend (gym_genesis/CubeStack-v0, the registry's default robot).

Behavioural contract restated from /root/reference/gym_genesis/tasks/so101/cube_stack_batch.py:25-226 and the scene
builder /root/reference/gym_genesis/tasks/utils.py:593-794:
  * scene: island slab, SO-101 x1.3 yawed 90 deg at (-0.5, 0, 0.7), five 4 cm cubes on the slab             (utils.py:714-789)
  * reset(): z = island_top + 0.021, all cube quats (0,0,0,1); per env, IN A PYTHON LOOP, scalar draws
    x1 ~ U(-0.3,-0.1), y1 ~ U(-0.1,0.1), x2, y2 likewise, rejected and re-drawn until |p1 - p2| >= 0.06; then per
    distractor xd ~ U(-0.35,0) (B), yd ~ U(-0.2,0.2) (B); arm at deg2rad(0,-177,165,72,-83,0), zero velocity,
    PD targets = home; ONE physics step                                                                       (:65-118)
  * step(a): targets <- a[:, :5], a[:, 5:]; one physics step                                                  (:135-141)
  * reward = float32(|cube1.xy - cube2.xy| < 0.05 and cube1.z - cube2.z > 0.03), a torch tensor               (:144-153)
  * obs: agent_pos (B,6) = so_101.get_qpos(); environment_state (B,14)                                        (:155-177)
The declared environment_state space says 10 (:16) although 14 values are returned; reproduced as declared.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from ...backend import models
from ..stack_common import StackTaskBase

AGENT_DIM = len(models.SO101_JOINTS)
ENV_DIM = 10  # declared (cube_stack_batch.py:16); get_obs() returns 14 columns


class CubeStackBatch(StackTaskBase):
    AGENT_DIM = AGENT_DIM
    ENV_DIM = ENV_DIM
    ROBOT_ROOT = "so101_base"
    JOINTS = models.SO101_JOINTS
    EEF_LINK = "gripper"

    # cameras as created (utils.py:668-696) and as posed per env in get_obs() (cube_stack_batch.py:185-196); this task
    # renders per env whatever camera_capture_mode says (:181-222)
    CAM_TOP = ((0.0, 0.0, 1.5), (0.0, 0.0, 0.0), 40.0)
    CAM_SIDE = ((1.0, 0.0, 0.5), (0.0, 0.0, 0.5), 30.0)
    CAM_WRIST = ((0.4, 0.0, 0.7), (0.0, 0.0, 1.0), 70.0)
    PER_ENV_TOP = ((-0.05, 0.0, 1.8), (-0.2, 0.0, 0.5))
    PER_ENV_SIDE = ((0.07, -1.0, 1.6), (-0.08, 0.0, 0.7))
    PIXELS_ALWAYS_PER_ENV = True

    def _wrist_camera(self):
        # camera frame = gripper rotation * Rx(-pi/2 + 0.8), at gripper pos + (0.09, 0, -0.08); an OpenGL camera looks
        # along its -z with +y up; the image is then rotated by 180 degrees (:199-213).  A camera rolled by 180 degrees about its
        # view axis (up -> -up) draws exactly that rotated image, so there is no flip of 236 MB afterwards; the view direction and
        # the up vector are the gripper's rotation of two constant vectors (one FK read, two cross products for both).
        c = self.__dict__.get("_wrist_const")
        if c is None:
            a = -math.pi / 2 + 0.8
            c = self._wrist_const = (torch.tensor([0.09, 0.0, -0.08], device=self.device),
                                     torch.tensor([[[0.0, math.sin(a), -math.cos(a)],       # -Rx[:, 2]: the view direction
                                                    [0.0, -math.cos(a), -math.sin(a)]]],   # -Rx[:, 1]: up, rolled by 180 degrees
                                                  device=self.device))
        xpos, xquat = self._mir.get_links()
        q = xquat[:, self.eef.idx, :]
        pos = xpos[:, self.eef.idx, :] + c[0]
        w, u = q[:, :1].unsqueeze(1), q[:, 1:].unsqueeze(1).expand(-1, 2, -1)
        v = c[1].expand_as(u)
        t = 2.0 * torch.cross(u, v, dim=2)
        d = v + w * t + torch.cross(u, t, dim=2)  # (B, 2, 3): q v q*
        return pos, pos + d[:, 0], d[:, 1].contiguous(), False

    def _scene_builder(self):
        return models.so101_cube_stack_scene()

    def _set_robot(self, view):
        self.so_101 = view
        self.motors_dof = np.arange(5)
        self.fingers_dof = np.array([5])

    def _home_qpos(self):
        return tuple(math.radians(d) for d in models.SO101_STACK_HOME_DEG)  # cube_stack_batch.py:106

    def sample_spawn(self) -> np.ndarray:
        Bg, r = self.global_num_envs, self._random
        z = self.island_top_z + 0.02 + 0.001
        # Rejection sampling, four scalar draws per attempt (:72-86): x1, y1, x2, y2, accepted when the two cubes are 6 cm apart.
        # An attempt consumes four doubles of the stream whatever its outcome, so attempt i IS draws 4 i .. 4 i + 3 and env e takes the
        # e-th accepted attempt: the attempts are drawn and judged as arrays, and the stream is then put where the scalar loop would
        # have left it (rewind, draw exactly what it consumed).  Same positions, same stream afterwards
        # (tests/test_host_cpu.py), 16 ms -> 0.3 ms per reset at 4096 envs.
        state = r.get_state()
        m = Bg + Bg // 4 + 64
        while True:
            u = r.random_sample(4 * m).reshape(m, 4)
            x1, y1 = -0.3 + (-0.1 - -0.3) * u[:, 0], -0.1 + (0.1 - -0.1) * u[:, 1]  # uniform(low, high) = low + (high - low) * sample
            x2, y2 = -0.3 + (-0.1 - -0.3) * u[:, 2], -0.1 + (0.1 - -0.1) * u[:, 3]
            ok = np.flatnonzero(np.power((x2 - x1) ** 2 + (y2 - y1) ** 2, 0.5) >= 0.06)
            if len(ok) >= Bg:
                break
            r.set_state(state)
            m *= 2
        ok = ok[:Bg]
        r.set_state(state)
        r.random_sample(4 * (int(ok[-1]) + 1))
        p1 = np.stack([x1[ok], y1[ok], np.full(Bg, z)], axis=1)
        p2 = np.stack([x2[ok], y2[ok], np.full(Bg, z)], axis=1)
        cols = [p1, p2]
        for _ in range(3):  # distractors (:97-103)
            x = r.uniform(-0.35, 0.0, size=(Bg,))
            y = r.uniform(-0.2, 0.2, size=(Bg,))
            cols.append(np.stack([x, y, np.full(Bg, z)], axis=1))
        return np.stack(cols, axis=1).astype(np.float32)
