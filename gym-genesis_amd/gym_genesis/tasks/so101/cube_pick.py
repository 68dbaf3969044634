"""SO-101 cube-pick task on the MI355X backend (BASELINE.json configs[3]).

Contract restated from /root/reference/gym_genesis/tasks/so101/cube_pick.py:18-158 and the scene
builder /root/reference/gym_genesis/tasks/utils.py:428-590:
  * kitchen island slab (top z = 0.70003), SO-101 x4 at (-0.5, 0, 0.7), 4 cm cube on the slab  (utils.py:543-587)
  * friction 5 on robot and cube, kp 1000 / kv 200 on the five arm dofs                   (cube_pick.py:38-42)
  * reset(): x ~ U(-0.32, -0.28), y ~ U(-0.05, 0.05), z = island_top + 0.021; cube quat (1,0,0,0);
    arm qpos = targets = 0; NO physics step                                                 (:58-83)
  * step(a): targets <- a[:5] (arm) and a[5:] (gripper); one physics step                  (:100-106)
  * reward = float32(cube_z > 0.1) -- with the cube resting on a 0.70 m slab this is 1 from the
    first step on; reproduced literally                                                     (:108-113)
  * obs: agent_pos = [eef pos3, eef quat4, gripper q1] (8 values although the declared space says
    6), environment_state = [cube pos3, quat4, eef-cube 3, dist 1] (11, declared 10)       (:115-132, :45-56)

Position taken on the reference's defect C-2: the reference class is unbatched although
the registry routes the batched key to it; here every quantity carries the env axis (B, .), and
B = 1 draws the same RNG stream as the reference.  The arm model is re-stated (see backend/models.py).
"""
from __future__ import annotations

import random
from typing import Optional, Tuple

import numpy as np
import torch

from ..._gym import spaces
from ...backend import models
from ...backend.lib import MirScene
from ..spawn_ahead import UniformBlocksAhead
from ..views import CameraView, EntityView, SceneView

AGENT_DIM = len(models.SO101_JOINTS)  # declared space shape, as in the reference (cube_pick.py:15)
ENV_DIM = 10                          # declared space shape (cube_pick.py:16)
AGENT_OBS, ENV_OBS = 8, 11            # what get_obs() actually returns


class CubePick:
    def __init__(self, enable_pixels, observation_height, observation_width, num_envs, env_spacing,
                 camera_capture_mode, strip_environment_state, shard: Optional[Tuple[int, int]] = None, exact_contacts: bool = True):
        # exact_contacts (not a reference kwarg): see tasks/franka/cube_pick.py
        self.exact_contacts = bool(exact_contacts)
        self.enable_pixels = enable_pixels
        self.observation_height = observation_height
        self.observation_width = observation_width
        self.camera_capture_mode = camera_capture_mode
        self.strip_environment_state = strip_environment_state
        self.global_num_envs = int(num_envs)
        rank, world = shard if shard is not None else (0, 1)
        self.shard_lo = self.global_num_envs * rank // world
        self.shard_hi = self.global_num_envs * (rank + 1) // world
        self.num_envs = self.shard_hi - self.shard_lo
        self._random = np.random.RandomState()
        self._ahead = UniformBlocksAhead([(-0.32, -0.28, self.global_num_envs), (-0.05, 0.05, self.global_num_envs)])  # so101/cube_pick.py:61-62
        if enable_pixels and camera_capture_mode not in ("per_env", "global"):
            raise ValueError(f"Unknown camera_capture_mode: {camera_capture_mode}")  # so101/cube_pick.py:154-155
        builder = models.so101_cube_pick_scene()
        self._builder = builder
        self._mir = MirScene(builder.build(), self.num_envs)
        self._mir.set_diag(False)  # solver diagnostics (16 B per env-step) are a debugging aid: _mir.set_diag(True) to read them
        if self.exact_contacts and self._mir.kernel == 16:
            self._mir.set_exact_contacts(True)
        self.device = self._mir.device
        self.island_top_z = models.ISLAND_TOP_Z
        self.scene = SceneView(self._mir, env_spacing=env_spacing, global_num_envs=self.global_num_envs, offset=self.shard_lo)
        self.so_101 = EntityView(self._mir, builder, root="so101_base", dof_names=models.SO101_JOINTS)
        self.cube = EntityView(self._mir, builder, root="cube", dof_names=())
        self.eef = self.so_101.get_link("gripper")
        if enable_pixels:
            # The reference reads self.cam here but its scene builder only creates cam_top / cam_side / cam_wrist
            # (utils.py:499-525; AttributeError).  Intent implemented: one camera posed like get_obs() poses it,
            # envs_offset[i] + (3.5, 0, 2.5) -> envs_offset[i] + (0, 0, 0.5) (so101/cube_pick.py:139-147), fov 30 as in the
            # Franka pick task this code was copied from (franka/cube_pick.py:56-63).
            self.cam = CameraView(self._mir, builder, self.scene, res=(observation_width, observation_height),
                                  pos=(3.5, 0.0, 2.5), lookat=(0, 0, 0.5), fov=30)
        self.motors_dof = np.arange(5)
        self.fingers_dof = np.array([5])
        box = lambda n: spaces.Box(low=-np.inf, high=np.inf, shape=(n,), dtype=np.float32)  # noqa: E731
        if enable_pixels:  # so101/cube_pick.py:45-50
            self.observation_space = spaces.Dict({"agent_pos": box(AGENT_DIM), "pixels": spaces.Box(
                low=0, high=255, shape=(observation_height, observation_width, 3), dtype=np.uint8)})
        else:
            self.observation_space = spaces.Dict({"agent_pos": box(AGENT_DIM), "environment_state": box(ENV_DIM)})
        self.action_space = spaces.Box(low=-1.0, high=1.0, shape=(AGENT_DIM,), dtype=np.float32)
        B, dev = self.num_envs, self.device
        self._zero = torch.zeros((B, AGENT_DIM), dtype=torch.float32, device=dev)
        self._quat = torch.tensor([[1.0, 0.0, 0.0, 0.0]], dtype=torch.float32, device=dev).repeat(B, 1)
        self._agent, self._envst = self._mir.empty(AGENT_OBS), self._mir.empty(ENV_OBS)
        self._reward, self._term = self._mir.empty(), self._mir.empty(dtype=torch.uint8)

    def get_cams(self):
        if not self.enable_pixels:
            raise ValueError("Cameras are not enabled. Set `enable_pixels=True` when creating the environment.")
        return self.cam

    def seed(self, seed):
        np.random.seed(seed)
        random.seed(seed)
        self._random = np.random.RandomState(seed)
        self._ahead.invalidate()
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)
        self.action_space.seed(seed)

    def sample_spawn(self) -> np.ndarray:
        Bg = self.global_num_envs
        x, y = self._ahead.take(self._random)  # (x block, then y block; possibly drawn ahead, tasks/spawn_ahead.py)
        z = np.full((Bg,), self.island_top_z + 0.02 + 0.001)
        out = np.stack([x, y, z], axis=1).astype(np.float32)
        self._ahead.start(self._random)
        return out

    def reset(self):
        pos = self._mir.staged(self.sample_spawn()[self.shard_lo:self.shard_hi])  # (pinned: the reset kernel reads it in place)
        self._mir.reset(pos, self._quat, self._zero)  # no scene.step() here (so101/cube_pick.py:81)
        if self.enable_pixels and getattr(self, "record_video", False):
            self.cam.start_recording()                # so101/cube_pick.py:83-84
        return self.get_obs()

    def step(self, action, host_terminated: bool = False):
        mir = self._mir
        if not isinstance(action, torch.Tensor):
            action = torch.as_tensor(np.asarray(action))
        self._agent, self._envst, self._reward, self._term = mir.step_fresh(action.reshape(self.num_envs, AGENT_DIM), AGENT_OBS, ENV_OBS,
                                                                            host_terminated=host_terminated)
        if host_terminated and self.exact_contacts and self.enable_pixels:
            # (exact contacts: the images must be drawn behind the launches of mir_step_end that step the deferred envs: see the Franka task)
            self._closed_term = mir.step_end()
        return None, self._reward, None, self._pack_obs()

    def step_begin(self, action):
        """step() whose launch also delivers `terminated` to the host (GenesisEnv.step collects it with step_end())."""
        return self.step(action, host_terminated=True)

    def step_end(self) -> np.ndarray:
        closed = self.__dict__.pop("_closed_term", None)
        return closed if closed is not None else self._mir.step_end()

    def make_fast_step(self):
        """The whole of GenesisEnv.step as one flat closure (tasks/fast_step.py)."""
        from ..fast_step import make_fast_step

        def coerce(action):
            if not isinstance(action, torch.Tensor):
                return np.asarray(action).reshape(self.num_envs, AGENT_DIM)  # (stays NumPy: staged in pinned memory by the step)
            return action.reshape(self.num_envs, AGENT_DIM)

        return make_fast_step(self, self._mir, AGENT_DIM, AGENT_OBS, ENV_OBS, coerce)

    def step_raw(self, action_dev: torch.Tensor) -> None:
        self._mir.step_fused(action_dev, self._agent, self._envst, self._reward, self._term)

    def compute_reward(self):
        self.get_obs()
        return self._reward

    def get_obs(self):
        self._agent, self._envst, self._reward, self._term = self._mir.get_obs()
        return self._pack_obs()

    def _pack_obs(self):
        obs = {"agent_pos": self._agent, "environment_state": self._envst}
        if self.enable_pixels:  # so101/cube_pick.py:134-157
            if self.strip_environment_state is True:
                del obs["environment_state"]
            obs["pixels"] = self.cam.render_envs() if self.camera_capture_mode == "per_env" else self.cam.render_global()
        return obs

    @property
    def terminated_device(self) -> torch.Tensor:
        return self._term
