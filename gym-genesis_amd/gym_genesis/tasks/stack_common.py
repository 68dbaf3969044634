"""Shared machinery of the two batched cube-stack tasks (gym_genesis/CubeStack-v0) on the MI355X backend.

The reference implements them as two near-identical classes
(/root/reference/gym_genesis/tasks/franka/cube_stack_kitchen_batch.py:27-224,
 /root/reference/gym_genesis/tasks/so101/cube_stack_batch.py:25-226); what differs between them -- robot, spawn
distributions and their RNG draw order, home pose, agent_pos layout, reward dtype -- stays in the subclasses, which
cite the reference line by line.  Everything here is plumbing: scene creation on the wave-per-env kernel, the fused
step, observation packing and the env-axis shard.
"""
from __future__ import annotations

import random
from typing import Optional, Tuple

import numpy as np
import torch

from .._gym import spaces
from ..backend import models
from ..backend.lib import MirScene
from .views import CameraView, EntityView, SceneView

ENV_OBS = 14  # [cube1 pos3, cube1 quat4, eef - cube1 3, |eef - cube1| 1, cube2 pos3]


class StackTaskBase:
    AGENT_DIM = 0          # declared action / agent_pos width
    ENV_DIM = ENV_OBS      # declared environment_state width
    ROBOT_ROOT = ""
    JOINTS: Tuple[str, ...] = ()
    EEF_LINK = ""

    def __init__(self, enable_pixels, observation_height, observation_width, num_envs, env_spacing, camera_capture_mode,
                 strip_environment_state, shard: Optional[Tuple[int, int]] = None):
        self.enable_pixels = enable_pixels
        self.observation_height = observation_height
        self.observation_width = observation_width
        self.camera_capture_mode = camera_capture_mode
        self.strip_environment_state = strip_environment_state
        self.env_spacing = env_spacing
        self.global_num_envs = int(num_envs)
        rank, world = shard if shard is not None else (0, 1)
        self.shard_lo = self.global_num_envs * rank // world
        self.shard_hi = self.global_num_envs * (rank + 1) // world
        self.num_envs = self.shard_hi - self.shard_lo
        self._random = np.random.RandomState()
        if enable_pixels and camera_capture_mode not in ("per_env", "global"):
            raise ValueError(f"Unknown camera_capture_mode: {camera_capture_mode}")
        builder = self._scene_builder()
        self._builder = builder
        self._mir = MirScene(builder.build(), self.num_envs)
        self._mir.set_diag(False)  # solver diagnostics (16 B per env-step) are a debugging aid: _mir.set_diag(True) to read them
        self.device = self._mir.device
        self.island_top_z = models.ISLAND_TOP_Z
        self.scene = SceneView(self._mir, env_spacing=env_spacing, global_num_envs=self.global_num_envs, offset=self.shard_lo)
        robot = EntityView(self._mir, builder, root=self.ROBOT_ROOT, dof_names=self.JOINTS)
        self._set_robot(robot)
        self.cube_1 = EntityView(self._mir, builder, root="cube_1", dof_names=())
        self.cube_2 = EntityView(self._mir, builder, root="cube_2", dof_names=())
        self.distractor_cubes = [EntityView(self._mir, builder, root=n, dof_names=()) for n in models.STACK_CUBES[2:]]
        self.eef = robot.get_link(self.EEF_LINK)
        if enable_pixels:  # utils.py:310-337 / :668-696 -- top, side (observation resolution) and wrist (always 640x480)
            res = (self.observation_width, self.observation_height)
            mk = lambda res_, cfg: CameraView(self._mir, builder, self.scene, res=res_, pos=cfg[0], lookat=cfg[1], fov=cfg[2])  # noqa: E731
            self.cam_top, self.cam_side = mk(res, self.CAM_TOP), mk(res, self.CAM_SIDE)
            self.cam_wrist = mk((640, 480), self.CAM_WRIST)
        self.observation_space = self._make_obs_space()
        self.action_space = spaces.Box(low=-1.0, high=1.0, shape=(self.AGENT_DIM,), dtype=np.float32)
        B, dev = self.num_envs, self.device
        self._home = torch.tensor(self._home_qpos(), dtype=torch.float32, device=dev).repeat(B, 1)
        self._quat = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=torch.float32, device=dev).repeat(B, len(models.STACK_CUBES), 1)
        self._agent, self._envst = self._mir.empty(self._mir.agent_dim), self._mir.empty(ENV_OBS)
        self._reward, self._term = self._mir.empty(), self._mir.empty(dtype=torch.uint8)

    # ---- provided by the subclasses -----------------------------------------------------------
    def _scene_builder(self):
        raise NotImplementedError

    def _set_robot(self, view):
        raise NotImplementedError

    def _home_qpos(self):
        raise NotImplementedError

    def sample_spawn(self) -> np.ndarray:
        """Cube spawn positions of the GLOBAL batch, float32 (B_global, 5, 3), drawn in the reference's order."""
        raise NotImplementedError

    # ---- reference surface ------------------------------------------------------------------------
    def _make_obs_space(self):
        box = lambda n: spaces.Box(low=-np.inf, high=np.inf, shape=(n,), dtype=np.float32)  # noqa: E731
        if self.enable_pixels:
            return spaces.Dict({
                "agent_pos": box(self.AGENT_DIM),
                "pixels": spaces.Box(low=0, high=255, shape=(self.observation_height, self.observation_width, 3), dtype=np.uint8),
            })
        return spaces.Dict({"agent_pos": box(self.AGENT_DIM), "environment_state": box(self.ENV_DIM)})

    def get_cams(self):
        if not self.enable_pixels:
            raise ValueError("Cameras are not enabled. Set `enable_pixels=True` when creating the environment.")
        return self.cam_top, self.cam_side, self.cam_wrist

    def seed(self, seed):
        np.random.seed(seed)
        random.seed(seed)
        self._random = np.random.RandomState(seed)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)
        self.action_space.seed(seed)

    def reset(self):
        pos = self._mir.staged(self.sample_spawn()[self.shard_lo:self.shard_hi])  # (pinned: the reset kernel reads it in place)
        self._mir.reset(pos, self._quat, self._home)  # set_pos/set_quat x5 + set_qpos(zero_velocity) + PD targets = home
        self._mir.step(1)                             # both references consume one physics step in reset()
        if self.enable_pixels and getattr(self, "record_video", False):  # cube_stack_kitchen_batch.py:110-113, so101/cube_stack_batch.py:114-117
            for cam in (self.cam_top, self.cam_side, self.cam_wrist):
                cam.start_recording()
        return self.get_obs()

    def reset_masked(self, env_mask):
        """Per-env reset without touching the other envs.  Draws one spawn per env from the task RandomState
        exactly like reset() (the host stream advances identically whether or not an env is selected); no physics step."""
        pos = self._mir.staged(self.sample_spawn()[self.shard_lo:self.shard_hi])  # (pinned: the reset kernel reads it in place)
        self._mir.reset(pos, self._quat, self._home, env_mask=env_mask)
        return self.get_obs()

    # ---- device-resident episode loop, same contract as FrankaCubePickBatch ----------------
    def enable_autoreset(self, max_episode_steps: int = 200, pool_len: int = 32):
        B, dev = self.num_envs, self.device
        pool = np.stack([self.sample_spawn()[self.shard_lo:self.shard_hi] for _ in range(pool_len)])  # (pool, B, 5, 3)
        self._spawn_pool = torch.from_numpy(pool).to(dev).contiguous()
        self._cursor = torch.zeros((B,), dtype=torch.int32, device=dev)
        self._episode_len = torch.zeros((B,), dtype=torch.int32, device=dev)
        self._truncated = torch.zeros((B,), dtype=torch.uint8, device=dev)
        self._done = torch.zeros((B,), dtype=torch.uint8, device=dev)
        self._max_episode_steps = int(max_episode_steps)

    def step_autoreset(self, action_dev: torch.Tensor):
        """One fused step, then device-side bookkeeping + re-spawn of finished envs (mir_autoreset); returns device tensors
        (agent_pos, environment_state, reward, terminated u8, truncated u8) of THIS step."""
        mir = self._mir
        mir.step_fused(action_dev, self._agent, self._envst, self._reward, self._term)
        mir.autoreset(self._term, self._episode_len, self._max_episode_steps, self._spawn_pool, self._cursor, self._quat, self._home,
                      self._truncated, self._done)
        return self._agent, self._envst, self._reward, self._term, self._truncated

    def rollout_autoreset(self, actions_dev: torch.Tensor, rows: torch.Tensor) -> torch.Tensor:
        """K steps of that loop in ONE launch: actions (K,B,n), rows (K,B,>= agent+14+3) <- [agent_pos | environment_state |
        reward | terminated | truncated] per step and env."""
        self._mir.rollout_autoreset(actions_dev, rows, self._episode_len, self._max_episode_steps, self._spawn_pool, self._cursor,
                                    self._quat, self._home)
        return rows

    def step(self, action, host_terminated: bool = False):
        mir = self._mir
        self._agent, self._envst, self._reward, self._term = mir.step_fresh(action, mir.agent_dim, ENV_OBS,
                                                                            host_terminated=host_terminated)
        return None, self._reward, None, self._pack_obs()

    def step_begin(self, action):
        """step() whose launch also delivers `terminated` to the host (GenesisEnv.step collects it with step_end())."""
        return self.step(action, host_terminated=True)

    def step_end(self):
        return self._mir.step_end()

    def make_fast_step(self):
        """The whole of GenesisEnv.step as one flat closure (tasks/fast_step.py)."""
        from .fast_step import make_fast_step
        return make_fast_step(self, self._mir, self.AGENT_DIM, self._mir.agent_dim, ENV_OBS)

    def step_raw(self, action_dev: torch.Tensor) -> None:
        self._mir.step_fused(action_dev, self._agent, self._envst, self._reward, self._term)

    def compute_reward(self):
        self.get_obs()
        return self._reward

    def get_obs(self):
        self._agent, self._envst, self._reward, self._term = self._mir.get_obs()
        return self._pack_obs()

    # cameras: (creation pos, creation lookat, fov) and the per-env poses of the reference's get_obs()
    CAM_TOP = CAM_SIDE = CAM_WRIST = ((0.0, 0.0, 1.0), (0.0, 0.0, 0.0), 40.0)
    PER_ENV_TOP = PER_ENV_SIDE = ((0.0, 0.0, 1.0), (0.0, 0.0, 0.0))
    PIXELS_ALWAYS_PER_ENV = False

    def _wrist_camera(self):
        """-> (pos (B,3), lookat (B,3), up (B,3) or None, rotate_180) of the link-mounted wrist camera."""
        raise NotImplementedError

    def _pixels(self):
        if self.PIXELS_ALWAYS_PER_ENV or self.camera_capture_mode == "per_env":
            pos, look, up, rot = self._wrist_camera()
            wrist = self.cam_wrist.render_cams(pos, look, up)
            if rot:
                wrist = torch.flip(wrist, dims=(1, 2))  # np.rot90(img, k=2)
            return {"top": self.cam_top.render_envs(*self.PER_ENV_TOP), "side": self.cam_side.render_envs(*self.PER_ENV_SIDE),
                    "wrist": wrist}
        if self.camera_capture_mode == "global":
            return {"top": self.cam_top.render_global(), "side": self.cam_side.render_global(), "wrist": self.cam_wrist.render_global()}
        raise ValueError(f"Unknown camera_capture_mode: {self.camera_capture_mode}")

    def _pack_obs(self):
        obs = {"agent_pos": self._agent, "environment_state": self._envst}
        if self.enable_pixels:
            if self.strip_environment_state:
                del obs["environment_state"]
            # dict of uint8 device tensors, (B,H,W,3) each (per_env) or (H,W,3) (global); the reference stacks host arrays
            obs["pixels"] = self._pixels()
        return obs

    @property
    def terminated_device(self) -> torch.Tensor:
        return self._term
