"""GenesisEnv — the gymnasium.Env adapter, backed by the MI355X rigid-body backend.

Contract restated from /root/reference/gym_genesis/env.py:13-125:
  ctor kwargs (17-29), ``metadata`` (15), ``reset`` -> (obs, {"is_success": [False]*B}) (48-57),
  ``step`` -> (obs, reward, terminated np.bool_ (B,), truncated zeros (B,), {"is_success"}) (61-69),
  ``render`` (97-98), ``push`` (59-60), accessors (84-95), task lookup by (robot, task, batched)
  raising NotImplementedError(key) for unknown combinations (100-125).

Positions taken on the reference's defects: ``terminated`` is derived
from the reward whatever its array type (C-1); ``get_robot``/``get_cube`` return the task's
robot / cube under whichever attribute name the task uses (C-3).
"""
from __future__ import annotations

import importlib
import warnings
from typing import Optional, Tuple

import numpy as np
import torch

from ._gym import Env

# (robot, task, batched) -> "module:Class" ; resolved lazily so importing gym_genesis stays light
_TASKS = {
    ("franka", "cube_pick", True): "gym_genesis.tasks.franka.cube_pick:FrankaCubePickBatch",
    ("so101", "cube_pick", True): "gym_genesis.tasks.so101.cube_pick:CubePick",
    ("so101", "cube_stack", True): "gym_genesis.tasks.so101.cube_stack_batch:CubeStackBatch",
    ("franka", "cube_stack", True): "gym_genesis.tasks.franka.cube_stack_kitchen_batch:FrankaCubeStackKitchenBatch",
    ("so101", "cube_stack", False): "gym_genesis.tasks.stack_one:CubeStackOne",            # num_envs = 0 (env.py:115)
    ("franka", "cube_stack", False): "gym_genesis.tasks.stack_one:FrankaCubeStackOne",     # num_envs = 0 (env.py:117)
}


class GenesisEnv(Env):
    metadata = {"render_modes": ["rgb_array"], "render_fps": 50}

    def __init__(self, task, robot="so101", enable_pixels=False, observation_height=480, observation_width=640,
                 num_envs=1, env_spacing=(1.0, 1.0), render_mode=None, camera_capture_mode="per_env",
                 strip_environment_state=True, shard: Optional[Tuple[int, int]] = None, record_video: bool = False, **task_kwargs):
        # (shard, record_video and task_kwargs are additions to the reference signature: the env-axis shard of this process; whether
        # reset() starts the camera recording that save_video() writes -- the reference always does with pixels, here it is asked
        # for, because the recorded frames stay alive and the README loop then runs 6-25 % slower; and options of this
        # backend's scene restatement, e.g. link_shape="capsule" for the Franka pick task)
        super().__init__()
        self.task = task
        self.robot = robot
        self.enable_pixels = enable_pixels
        self.observation_height = observation_height
        self.observation_width = observation_width
        self.env_spacing = env_spacing
        self.render_mode = render_mode
        self.camera_capture_mode = camera_capture_mode
        self.strip_environment_state = strip_environment_state
        self._shard = shard
        self._task_kwargs = task_kwargs
        self.num_envs = num_envs
        self._env = self._make_env_task(task)
        self._env.record_video = bool(record_video)
        # local shard size when sharded; the unbatched tasks (num_envs = 0) keep 0 like the reference (env.py:56,65)
        self.num_envs = 0 if getattr(self._env, "unbatched", False) else self._env.num_envs
        self.observation_space = self._env.observation_space
        self.action_space = self._env.action_space
        self.scene = None
        # the two halves of the fast step path, looked up once (they sit in front of every launch, where the GPU idles)
        # (with pixels too: the renders of the observation are queued behind the step launch, and step_end() waits for that launch's
        #  terminated bytes only -- not, as a `.cpu()` of the mask would, for the images)
        fast = hasattr(self._env, "step_begin") and not getattr(self._env, "unbatched", False)
        self._begin = self._env.step_begin if fast else None
        self._end = self._env.step_end if fast else None
        if fast and not self.enable_pixels and hasattr(self._env, "make_fast_step"):
            # the state-only batched tasks supply the whole of step() as one flat function; bound on the instance, so
            # `env.step(a)` calls it without passing through this class's method
            self.step = self._env.make_fast_step()

    # ---- gymnasium API ------------------------------------------------------------------------
    def reset(self, seed=None, options=None):
        super().reset(seed=seed)
        if seed is not None:
            self._env.seed(seed)
        observation = self._env.reset()
        return observation, {"is_success": [False] * self.num_envs}

    def step(self, action):
        begin = self._begin
        if begin is not None:
            # Fast path: the launch stores `terminated` into pinned host memory itself.  Everything the API returns besides
            # that mask is built while the kernel runs; step_end() then waits for the launch and hands over a fresh NumPy
            # bool array -- the reference's `is_success.detach().cpu().numpy().astype(bool)` (env.py:64).
            _, reward, _, observation = begin(action)
            is_success = self._env.terminated_device.view(torch.bool)
            truncated = np.zeros(self.num_envs, dtype=bool)
            info = {"is_success": is_success}
            terminated = self._end()
            return observation, reward, terminated, truncated, info
        _, reward, _, observation = self._env.step(action)
        term_dev = getattr(self._env, "terminated_device", None)
        if term_dev is not None and not getattr(self._env, "unbatched", False):
            is_success = term_dev.view(torch.bool)         # 0/1 bytes written by the same kernel as the reward (no extra launch)
        elif isinstance(reward, torch.Tensor):
            is_success = reward == 1
        else:
            is_success = torch.as_tensor(np.asarray(reward) == 1)
        terminated = is_success.detach().cpu().numpy()     # the one D->H sync the API mandates (a fresh bool array)
        truncated = np.zeros(self.num_envs, dtype=bool)
        return observation, reward, terminated, truncated, {"is_success": is_success}

    def render(self):
        return self._env.cam.render()[0] if self.enable_pixels else None

    def close(self):
        pass

    # ---- extras the reference exposes ---------------------------------------------------------
    def push(self):
        self._env.scene.step()

    def save_video(self, save_video: bool = False, file_name: str = "episode.mp4", fps=60):
        if self.enable_pixels and save_video:
            warnings.warn("Calling `save_video()` stops the camera recording; no further frames can be recorded.",
                          stacklevel=2)
            # (the three-camera stack tasks have no `cam`: the reference raises AttributeError there; the top view is saved here)
            cam = getattr(self._env, "cam", None) or self._env.cam_top
            if not getattr(cam, "recording", False):
                warnings.warn(f"no recording is running (GenesisEnv(..., record_video=True) starts one at every reset): nothing was "
                              f"written to {file_name!r}", stacklevel=2)
                return
            cam.stop_recording(save_to_filename=file_name, fps=fps)

    def get_obs(self):
        return self._env.get_obs()

    def get_cams(self):
        return self._env.get_cams()

    def get_cube(self):
        for name in ("cube_1", "cube"):
            if hasattr(self._env, name):
                return getattr(self._env, name)
        raise AttributeError("task has no cube")

    def get_robot(self):
        for name in ("so_101", "franka"):
            if hasattr(self._env, name):
                return getattr(self._env, name)
        raise AttributeError("task has no robot")

    # ---- task factory -------------------------------------------------------------------------
    def _make_env_task(self, task_name):
        key = (self.robot, task_name, self.num_envs > 0)
        if key not in _TASKS:
            raise NotImplementedError(key)
        mod, cls = _TASKS[key].split(":")
        ctor = getattr(importlib.import_module(mod), cls)
        kwargs = dict(
            enable_pixels=self.enable_pixels,
            observation_height=self.observation_height,
            observation_width=self.observation_width,
            num_envs=self.num_envs,
            env_spacing=self.env_spacing,
            camera_capture_mode=self.camera_capture_mode,
            strip_environment_state=self.strip_environment_state,
        )
        if self._shard is not None:
            kwargs["shard"] = self._shard
        kwargs.update(self._task_kwargs)
        return ctor(**kwargs)
