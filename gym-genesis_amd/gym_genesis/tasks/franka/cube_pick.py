"""Batched Franka cube-pick task on the MI355X backend.

Behavioural contract restated from the reference task
(/root/reference/gym_genesis/tasks/franka/cube_pick.py:21-181):
  * scene: plane + Panda (9 dofs) + 4 cm cube at (0.65, 0, 0.02), dt = 0.01      (:37-54)
  * reset(): cube x ~ U(0.45, 0.80), y ~ U(-0.25, 0.25) from the task RandomState, drawn x
    first then y, z = 0.02; cube quaternion (0,0,0,1) wxyz; arm at the home pose with zero
    velocity; PD targets = home; ONE physics step; returns get_obs()               (:86-112)
  * step(a): PD targets <- a[:, :7] (rad) and a[:, 7:] (m), unscaled and unclipped; one
    physics step; returns (None, reward, None, obs)                                (:122-128)
  * reward = float32(cube_z > 0.1)                                                 (:130-135)
  * obs: agent_pos (B,9) = [eef pos3, eef quat4 wxyz, finger q2]; environment_state (B,11) =
    [cube pos3, cube quat4, eef - cube 3, |eef - cube| 1]; torch float32 on device (:137-152)

Differences by design: control + step + reward + observation extraction are ONE kernel
launch (mir_step_fused) instead of ~20 launches and two host syncs, and ``reward`` is a
torch tensor on the device (its values equal the reference's NumPy array).
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

AGENT_DIM = len(models.FRANKA_JOINTS)
ENV_DIM = 11


class FrankaCubePickBatch:
    def __init__(self, enable_pixels, observation_height, observation_width, num_envs, env_spacing,
                 camera_capture_mode, strip_environment_state, shard: Optional[Tuple[int, int]] = None, link_shape: str = "capsule",
                 contact_capacity: int = 16, exact_contacts: bool = True):
        # link_shape: collision stand-ins of links 1-7, "box" or "capsule" (models._add_franka); not a reference kwarg
        # contact_capacity: contact points kept per env (not a reference kwarg; Genesis keeps 100+ pairs).  16 = the 16-lane kernel
        # (manifolds thinned beyond that: 29 % of the env-steps of the reference's expert, same success rate -- tests/test_ref_expert.py);
        # 17 .. 48 = the same scene on the wave-per-env kernel, never thinned by that policy, several times slower
        # exact_contacts (not a reference kwarg; DEFAULT ON since round 6): Genesis keeps every contact point of its candidate pairs
        # (RigidOptions at /root/reference/gym_genesis/tasks/franka/cube_pick.py:46).  The 16-lane kernel serves the envs with at most 16
        # points exactly; the envs whose narrowphase finds more -- only those, step by step -- are stepped with 48 points, never thinned,
        # by its three-contacts-per-lane instantiation.  A workload that never exceeds 16 points (the headline's random
        # targets) computes the same bits at the same speed with the switch on or off.  exact_contacts=False is the SPEED KNOB: manifolds
        # thinned to 16 points (oracle/orc_rigid.c: thin_manifolds), e.g. 48 instead of ~100 us per env.step on the reference's expert
        # at 4096 envs, same success rate (tests/test_ref_expert.py) but other trajectories from the first thinned step on.
        # reset() / step() / env.step() only; the device-side episode loops (rollout_autoreset, step_packed) have no host in them to
        # close a step on and are refused by the library while the switch is on (set_exact_contacts(False) first).
        self.contact_capacity = int(contact_capacity)
        self.exact_contacts = bool(exact_contacts)
        self.enable_pixels = enable_pixels
        self.observation_height = observation_height
        self.observation_width = observation_width
        self.camera_capture_mode = camera_capture_mode
        self.strip_environment_state = strip_environment_state
        self.env_spacing = env_spacing
        self.link_shape = link_shape
        # env-axis shard: this process owns envs [lo, hi) of the global batch
        self.global_num_envs = int(num_envs)
        rank, world = shard if shard is not None else (0, 1)
        self.shard_lo = self.global_num_envs * rank // world
        self.shard_hi = self.global_num_envs * (rank + 1) // world
        self.num_envs = self.shard_hi - self.shard_lo
        self._random = np.random.RandomState()
        # the next reset's draws (x block, then y block: cube_pick.py:90-91), made a chunk per env.step() while its kernel runs
        self._ahead = UniformBlocksAhead([(0.45, 0.80, self.global_num_envs), (-0.25, 0.25, self.global_num_envs)])
        self._build_scene()
        self.observation_space = self._make_obs_space()
        self.action_space = spaces.Box(low=-1.0, high=1.0, shape=(AGENT_DIM,), dtype=np.float32)

    # ---- scene ------------------------------------------------------------------------------
    def _build_scene(self):
        if self.enable_pixels and self.camera_capture_mode not in ("per_env", "global"):
            raise ValueError(f"Unknown camera_capture_mode: {self.camera_capture_mode}")  # cube_pick.py:177-178
        builder = models.franka_cube_pick_scene(link_shape=self.link_shape)
        if self.contact_capacity != 16:
            builder.opt["max_contacts"] = self.contact_capacity
        self._builder = builder
        self._mir = MirScene(builder.build(), self.num_envs)
        self._mir.set_diag(False)  # solver diagnostics (16 B per env-step) are a debugging aid: _mir.set_diag(True) to read them
        if self.exact_contacts and self._mir.kernel == 16:
            self._mir.set_exact_contacts(True)
        B, dev = self.num_envs, self._mir.device
        self.device = dev
        self.scene = SceneView(self._mir, env_spacing=self.env_spacing, global_num_envs=self.global_num_envs,
                               offset=self.shard_lo)
        self.franka = EntityView(self._mir, builder, root="link0", dof_names=models.FRANKA_JOINTS)
        self.cube = EntityView(self._mir, builder, root="cube", dof_names=())
        self.eef = self.franka.get_link("hand")
        if self.enable_pixels:  # cube_pick.py:55-63
            self.cam = CameraView(self._mir, builder, self.scene, res=(self.observation_width, self.observation_height),
                                  pos=(3.5, 0.0, 2.5), lookat=(0, 0, 0.5), fov=30)
        self.motors_dof = np.arange(7)
        self.fingers_dof = np.arange(7, 9)
        # persistent device buffers: one set, rewritten by every fused step
        self._agent = torch.empty((B, AGENT_DIM), dtype=torch.float32, device=dev)
        self._envst = torch.empty((B, ENV_DIM), dtype=torch.float32, device=dev)
        self._reward = torch.empty((B,), dtype=torch.float32, device=dev)
        self._term = torch.empty((B,), dtype=torch.uint8, device=dev)
        self._home = torch.tensor(models.FRANKA_HOME, dtype=torch.float32, device=dev).repeat(B, 1)
        self._quat = torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=torch.float32, device=dev).repeat(B, 1)

    def _make_obs_space(self):
        box = lambda n: spaces.Box(low=-np.inf, high=np.inf, shape=(n,), dtype=np.float32)  # noqa: E731
        if self.enable_pixels:
            return spaces.Dict({
                "agent_pos": box(AGENT_DIM),
                "pixels": spaces.Box(low=0, high=255, shape=(self.observation_height, self.observation_width, 3),
                                     dtype=np.uint8),
            })
        return spaces.Dict({"agent_pos": box(AGENT_DIM), "environment_state": box(ENV_DIM)})

    def get_cams(self):
        if not self.enable_pixels:
            raise ValueError("Cameras are not enabled. Set `enable_pixels=True` when creating the environment.")
        return self.cam

    # ---- episode control --------------------------------------------------------------------
    def seed(self, seed):
        np.random.seed(seed)
        random.seed(seed)
        self._random = np.random.RandomState(seed)
        self._ahead.invalidate()  # (draws made ahead came from the old stream)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)
        self.action_space.seed(seed)

    def sample_spawn(self) -> np.ndarray:
        """Cube spawn positions for the GLOBAL batch (so a sharded run draws the same stream as an
        unsharded one), float32 (B_global, 3).  The draws may have been made ahead of time (tasks/spawn_ahead.py): same
        stream, same values; the next block is then started."""
        Bg = self.global_num_envs
        x, y = self._ahead.take(self._random)
        out = np.empty((Bg, 3), dtype=np.float32)  # (x block, then y block, as the reference draws them; rounded to float32 once)
        out[:, 0] = x
        out[:, 1] = y
        out[:, 2] = 0.02
        self._ahead.start(self._random)
        return out

    def reset(self):
        pos = self._mir.staged(self.sample_spawn()[self.shard_lo:self.shard_hi])  # (pinned: the reset kernel reads it in place)
        self._mir.reset(pos, self._quat, self._home)  # set_pos/set_quat/set_qpos(zero_velocity) + PD targets = home
        self._mir.step(1)                             # the reference consumes one physics step in reset()
        if self.enable_pixels and getattr(self, "record_video", False):
            self.cam.start_recording()                # cube_pick.py:109-110
        return self.get_obs()

    def reset_masked(self, env_mask):
        """Per-env reset without touching the other envs (the reference can only reset
        all envs at once, README.md:41-43).  `env_mask`: (B,) bool/uint8 tensor or array, device or host.
        Draws one spawn per env from the task RandomState exactly like reset() (so the host stream
        advances identically whether or not an env is selected); masked envs get the home pose, zero
        velocity and PD targets = home.  No physics step is consumed: the other envs must not advance."""
        pos = self._mir.staged(self.sample_spawn()[self.shard_lo:self.shard_hi])  # (pinned: the reset kernel reads it in place)
        self._mir.reset(pos, self._quat, self._home, env_mask=env_mask)
        return self.get_obs()

    # ---- device-resident episode loop --------------------------------------
    def enable_autoreset(self, max_episode_steps: int = 200, pool_len: int = 64):
        """Keep the whole episode loop on the device: after every step_autoreset() the envs whose episode
        ended (terminated, or `max_episode_steps` reached) are re-spawned by one small kernel, without the
        D->H read of `terminated` and the reset-all the reference's loop needs (README.md:41-43).
        Spawn positions come from the task RandomState as in reset() (x block then y block per draw),
        `pool_len` draws ahead; env e's k-th re-spawn uses draw k % pool_len."""
        B, dev = self.num_envs, self.device
        if getattr(self._mir, "exact_contacts", False):
            # (exact contacts close every step on the host -- that is where the envs with more than 16 contact points are handed on;
            #  this loop has no host in it: it runs with the manifolds thinned to the 16-lane kernel's 16 points)
            import warnings

            warnings.warn("enable_autoreset(): the device-resident episode loop runs with exact_contacts switched off (contact manifolds "
                          "thinned to 16 points per env); GenesisEnv(..., exact_contacts=False) says so up front", stacklevel=2)
            self._mir.set_exact_contacts(False)
            self.exact_contacts = False
        pool = np.stack([self.sample_spawn()[self.shard_lo:self.shard_hi] for _ in range(pool_len)])
        self._spawn_pool = torch.from_numpy(pool).to(dev).contiguous()
        self._cursor = torch.zeros((B,), dtype=torch.int32, device=dev)
        self._episode_len = torch.zeros((B,), dtype=torch.int32, device=dev)
        self._truncated = torch.zeros((B,), dtype=torch.uint8, device=dev)
        self._done = torch.zeros((B,), dtype=torch.uint8, device=dev)
        self._max_episode_steps = int(max_episode_steps)

    def step_autoreset(self, action_dev: torch.Tensor):
        """One fused step, then device-side bookkeeping.  Returns device tensors (agent_pos, environment_state,
        reward, terminated u8, truncated u8) of THIS step (the terminal observation for envs that just ended);
        those envs start their next step from the re-spawned state, whose first observation is therefore one
        physics step after the spawn, as after reset() (cube_pick.py:107)."""
        mir = self._mir
        mir.step_fused(action_dev, self._agent, self._envst, self._reward, self._term)
        mir.autoreset(self._term, self._episode_len, self._max_episode_steps, self._spawn_pool, self._cursor,
                      self._quat, self._home, self._truncated, self._done)
        return self._agent, self._envst, self._reward, self._term, self._truncated

    def rollout_autoreset(self, actions_dev: torch.Tensor, rows: torch.Tensor) -> torch.Tensor:
        """K steps of the device-resident episode loop in ONE launch (after enable_autoreset()): actions (K,B,9), rows
        (K,B,>=23) <- [agent_pos 9 | environment_state 11 | reward | terminated | truncated] per step and env."""
        self._mir.rollout_autoreset(actions_dev, rows, self._episode_len, self._max_episode_steps, self._spawn_pool, self._cursor,
                                    self._quat, self._home)
        return rows

    def step(self, action):
        # fresh output tensors per call, like the reference (callers may keep old observations)
        mir = self._mir
        self._agent, self._envst, self._reward, self._term = mir.step_fresh(action, AGENT_DIM, ENV_DIM)
        return None, self._reward, None, self._pack_obs()

    def step_begin(self, action):
        """step() for GenesisEnv.step: the same launch also delivers `terminated` to the host; GenesisEnv prepares its other
        return values while the kernel runs and then collects the mask with step_end()."""
        mir = self._mir
        self._agent, self._envst, self._reward, self._term = mir.step_fresh(action, AGENT_DIM, ENV_DIM,
                                                                            host_terminated=True)
        if self.exact_contacts and self.enable_pixels:
            # (exact contacts: a deferred env is stepped by the launches of mir_step_end -- the images of the observation, drawn by
            #  _pack_obs below, must come behind them: the step is closed first and step_end() hands the mask over)
            self._closed_term = mir.step_end()
        return None, self._reward, None, self._pack_obs()

    def step_end(self) -> np.ndarray:
        closed = self.__dict__.pop("_closed_term", None)
        return closed if closed is not None else self._mir.step_end()

    def make_fast_step(self):
        """The whole of GenesisEnv.step as one flat closure (tasks/fast_step.py)."""
        from ..fast_step import make_fast_step
        return make_fast_step(self, self._mir, AGENT_DIM, AGENT_DIM, ENV_DIM)

    def step_raw(self, action_dev: torch.Tensor) -> None:
        """Hot path without any Python-side packing: `action_dev` is a contiguous float32 (B,9) device tensor."""
        self._mir.step_fused(action_dev, self._agent, self._envst, self._reward, self._term)

    def compute_reward(self):
        self._refresh()
        return self._reward

    def get_obs(self):
        self._refresh()
        return self._pack_obs()

    # ---- helpers ----------------------------------------------------------------------------
    def _as_action(self, action) -> torch.Tensor:
        return self._mir.as_action(action, AGENT_DIM)

    def _refresh(self):
        agent, env, rew, term = self._mir.get_obs()
        self._agent, self._envst, self._reward, self._term = agent, env, rew, term

    def _pack_obs(self):
        obs = {"agent_pos": self._agent, "environment_state": self._envst}
        if self.enable_pixels:  # cube_pick.py:159-180
            if self.strip_environment_state is True:
                del obs["environment_state"]
            if self.camera_capture_mode == "per_env":
                # one batched launch: env i alone, camera at envs_offset[i] + (3.5, 0, 2.5) looking at
                # envs_offset[i] + (0, 0, 0.5) == the fixed pose in env i's own frame.  uint8 (B, H, W, 3) ON DEVICE
                # (the reference stacks B host arrays; np.asarray(pixels.cpu()) has its layout and dtype).
                pixels = self.cam.render_envs()
            elif self.camera_capture_mode == "global":
                pixels = self.cam.render_global()  # (H, W, 3)
            else:
                raise ValueError(f"Unknown camera_capture_mode: {self.camera_capture_mode}")
            obs["pixels"] = pixels
        return obs

    @property
    def terminated_device(self) -> torch.Tensor:
        return self._term
