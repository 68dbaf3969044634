"""CPU-tier tests of the cube-stack tasks (gym_genesis/CubeStack-v0): reset RNG streams against the golden fixture
restated from the reference's draw order, the drop-in contract of both task classes on the oracle-backed test double,
and known-answer physics of the stack scenes on the float64 oracle (resting cubes, a stacked cube, the reward rule)."""
import json
import math
import os

import numpy as np
import pytest
import torch

import orc
from gym_genesis.backend import models

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _task(monkeypatch, robot, B):
    import fake_scene
    from gym_genesis.tasks import stack_common

    monkeypatch.setattr(stack_common, "MirScene", fake_scene.OracleScene)
    if robot == "franka":
        from gym_genesis.tasks.franka.cube_stack_kitchen_batch import FrankaCubeStackKitchenBatch as T
    else:
        from gym_genesis.tasks.so101.cube_stack_batch import CubeStackBatch as T
    return T(False, 480, 640, B, (1.0, 1.0), "global", True)


def test_stack_reset_rng_streams_match_golden_fixture(monkeypatch):
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "stack_reset_rng.json")))
    for key, g in gold.items():
        task = _task(monkeypatch, g["robot"], g["B"])
        task.seed(g["seed"])
        pos = task.sample_spawn()
        assert pos.shape == (g["B"], 5, 3) and pos.dtype == np.float32
        assert np.array_equal(pos[0], np.array(g["first_env_f32"], np.float32)), key
        assert np.array_equal(pos[-1], np.array(g["last_env_f32"], np.float32)), key
        assert float(pos.astype(np.float64).sum()) == g["sum_f64"]
        assert task._random.uniform() == g["next_uniform_f64"]  # the stream was consumed exactly as far as the reference does


@pytest.mark.parametrize("robot,agent_dim,declared_env", [("franka", 9, 14), ("so101", 6, 10)])
def test_stack_env_contract(monkeypatch, robot, agent_dim, declared_env):
    import fake_scene
    from gym_genesis.env import GenesisEnv
    from gym_genesis.tasks import stack_common

    monkeypatch.setattr(stack_common, "MirScene", fake_scene.OracleScene)
    B = 2
    env = GenesisEnv(task="cube_stack", robot=robot, num_envs=B, enable_pixels=False)
    obs, info = env.reset(seed=5)
    assert info == {"is_success": [False] * B}
    assert obs["agent_pos"].shape == (B, agent_dim) and obs["environment_state"].shape == (B, 14)
    assert env.observation_space["agent_pos"].shape == (agent_dim,) and env.observation_space["environment_state"].shape == (declared_env,)
    assert env.action_space.shape == (agent_dim,)
    es = obs["environment_state"].numpy()
    task = env._env
    # cube_1 / cube_2 where the seeded stream put them, on the slab, quaternion (0,0,0,1) as set (…:68,94 / :67)
    task.seed(5)
    spawn = task.sample_spawn()
    # (randomly spawned cubes may overlap a distractor and get pushed apart by the contact during reset()'s one step)
    assert np.allclose(es[:, :2], spawn[:, 0, :2], atol=5e-3) and np.allclose(es[:, 11:13], spawn[:, 1, :2], atol=5e-3)
    assert (np.abs(es[:, 2] - (models.ISLAND_TOP_Z + 0.02)) < 2e-3).all() and np.allclose(es[:, 3:7], [0, 0, 0, 1], atol=1e-5)
    assert np.allclose(es[:, 10], np.linalg.norm(es[:, 7:10], axis=1), atol=1e-6)
    if robot == "so101":  # agent_pos = qpos, at the home pose after one step
        assert np.allclose(obs["agent_pos"].numpy(), np.radians(models.SO101_STACK_HOME_DEG), atol=2e-2)
    else:
        assert np.allclose(es[:, 7:10], obs["agent_pos"].numpy()[:, :3] - es[:, :3], atol=1e-6)
    act = np.tile(np.asarray(task._home_qpos(), np.float32), (B, 1))
    obs2, reward, terminated, truncated, info = env.step(act)
    assert reward.shape == (B,) and reward.dtype == torch.float32 and not terminated.any() and not truncated.any()
    assert env.get_cube() is task.cube_1 and env.get_robot() is (task.franka if robot == "franka" else task.so_101)
    assert len(task.distractor_cubes) == 3 and task.cube_2.get_pos().shape == (B, 3)
    with pytest.raises(ValueError):
        env.get_cams()


def test_registry_routes_cube_stack(monkeypatch):
    import fake_scene
    import gym_genesis  # noqa: F401
    from gym_genesis import _gym
    from gym_genesis.tasks import stack_common

    monkeypatch.setattr(stack_common, "MirScene", fake_scene.OracleScene)
    env = _gym.make("gym_genesis/CubeStack-v0", num_envs=1)  # registry default robot = so101 (__init__.py:26)
    inner = env.unwrapped if hasattr(env, "unwrapped") else env
    assert type(inner._env).__name__ == "CubeStackBatch"


def _settled_oracle(builder, pos, steps=150):
    spec = builder.build()
    o = orc.Oracle(spec, 1)
    home = np.asarray(models.FRANKA_HOME, dtype=np.float64)
    o.reset(np.asarray(pos, np.float64)[None], np.tile([0, 0, 0, 1.0], (1, 5, 1)), home[None])
    for _ in range(steps):
        o.step()
    return o


def test_five_cubes_rest_on_the_slab_and_a_stacked_cube_flips_the_reward():
    b = models.franka_cube_stack_scene()
    z = models.STACK_CUBE_Z
    far = [(0.2, -0.15, z), (-0.2, -0.2, z), (0.05, 0.2, z)]
    # cube_1 next to cube_2: no reward; five resting cubes = 20 plane-box contacts, no motion
    o = _settled_oracle(b, [(0.1, 0.0, z), (-0.1, 0.05, z)] + far)
    ncon, nefc, _ = o.counts()
    assert ncon == 20
    _, env, rew, term = o.get_obs()
    assert rew[0] == 0.0 and not term[0]
    assert abs(env[0, 2] - (models.ISLAND_TOP_Z + 0.02)) < 1e-3 and np.abs(o.state()[1][0, 9:]).max() < 1e-3
    # force balance on a resting cube: total normal force = m g (soft-constraint steady state)
    m = 200.0 * 0.04 ** 3
    f = o.read(orc.F_EFCFORCE)
    # cube_1 on top of cube_2: box-box face contact holds it, reward = 1 (xy distance 0, z difference 0.04 > 0.03)
    o = _settled_oracle(b, [(-0.1, 0.05, z + 0.0405), (-0.1, 0.05, z)] + far)
    _, env, rew, term = o.get_obs()
    assert rew[0] == 1.0 and term[0]
    assert abs((env[0, 2] - env[0, 13]) - 0.04) < 2e-3 and np.hypot(env[0, 0] - env[0, 11], env[0, 1] - env[0, 12]) < 1e-3
    assert np.abs(o.state()[1][0, 9:]).max() < 5e-3
    assert f.shape[0] >= 80 and m > 0
    # just outside the xy tolerance: stacked but 0.051 m off -> the top cube tips off or stays, either way no reward while offset
    spec = b.build()
    o2 = orc.Oracle(spec, 1)
    o2.reset(np.array([[(-0.1 + 0.051, 0.05, z + 0.0405), (-0.1, 0.05, z)] + far]), np.tile([0, 0, 0, 1.0], (1, 5, 1)),
             np.asarray(models.FRANKA_HOME)[None])
    o2.fk()
    assert o2.get_obs()[2][0] == 0.0  # reward rule is strict on the xy distance (cube_stack_kitchen_batch.py:144)


def test_so101_stack_home_pose_is_clear_of_the_slab():
    b = models.so101_cube_stack_scene()
    spec = b.build()
    o = orc.Oracle(spec, 1)
    z = models.STACK_CUBE_Z
    pos = [(x, y, z) for x, y in models._STACK_CUBE_XY]
    home = np.radians(models.SO101_STACK_HOME_DEG)
    o.reset(np.array([pos]), np.tile([0, 0, 0, 1.0], (1, 5, 1)), home[None])
    for _ in range(100):
        o.step()
    assert o.counts()[0] == 20  # only the cubes touch the slab
    agent, _, _, _ = o.get_obs()
    assert np.abs(agent[0] - home).max() < 5e-3 and math.isfinite(agent.sum())


@pytest.mark.parametrize("robot,mode", [("franka", "per_env"), ("franka", "global"), ("so101", "global")])
def test_stack_pixels_contract_on_the_test_double(monkeypatch, robot, mode):
    """enable_pixels=True: obs["pixels"] is a dict of the three views (cube_stack_kitchen_batch.py:169-222,
    so101/cube_stack_batch.py:181-224); wrist camera fixed at 640x480 (utils.py:331,688); SO-101 renders per env whatever
    the capture mode."""
    import fake_scene
    from gym_genesis.env import GenesisEnv
    from gym_genesis.tasks import stack_common
    from gym_genesis.tasks import views

    monkeypatch.setattr(stack_common, "MirScene", fake_scene.OracleScene)
    B, H, W = 2, 12, 16
    env = GenesisEnv(task="cube_stack", robot=robot, num_envs=B, enable_pixels=True, observation_height=H, observation_width=W,
                     camera_capture_mode=mode)
    task = env._env
    task.cam_wrist.res = (32, 24)  # keep the float64 ray caster quick in the CPU tier (the product keeps 640x480)
    obs, _ = env.reset(seed=1)
    assert set(obs) == {"agent_pos", "pixels"} and set(obs["pixels"]) == {"top", "side", "wrist"}
    per_env = mode == "per_env" or robot == "so101"
    lead = (B,) if per_env else ()
    assert tuple(obs["pixels"]["top"].shape) == lead + (H, W, 3) and tuple(obs["pixels"]["side"].shape) == lead + (H, W, 3)
    assert tuple(obs["pixels"]["wrist"].shape) == lead + (24, 32, 3) and obs["pixels"]["wrist"].dtype == torch.uint8
    assert env.get_cams() == (task.cam_top, task.cam_side, task.cam_wrist)
    assert task.cam_top.fov == 40.0 and task.cam_wrist.fov == (90.0 if robot == "franka" else 70.0)
    assert len(np.unique(obs["pixels"]["top"].numpy().reshape(-1, 3), axis=0)) >= 3  # slab, floor, cubes in the top view


@pytest.mark.parametrize("robot,n", [("franka", 9), ("so101", 6)])
def test_unbatched_stack_tasks(monkeypatch, robot, n):
    """num_envs = 0 routes to the unbatched classes (env.py:110-117): no batch axis, scalar reward, the batched class's
    B = 1 spawn stream (the reference draws the same values as scalars: cube_stack_one.py:68-86, so101/cube_stack.py:66-93)."""
    import fake_scene
    from gym_genesis.env import GenesisEnv
    from gym_genesis.tasks import stack_common

    monkeypatch.setattr(stack_common, "MirScene", fake_scene.OracleScene)
    env = GenesisEnv(task="cube_stack", robot=robot, num_envs=0, enable_pixels=False)
    assert type(env._env).__name__ == ("FrankaCubeStackOne" if robot == "franka" else "CubeStackOne") and env.num_envs == 0
    obs, info = env.reset(seed=2)
    assert info == {"is_success": []} and obs["agent_pos"].shape == (n,) and obs["environment_state"].shape == (14,)
    obs, reward, terminated, truncated, info = env.step(np.asarray(env._env._home_qpos(), np.float32))
    assert reward.dim() == 0 and terminated.shape == () and truncated.shape == (0,) and obs["agent_pos"].shape == (n,)
    ref = _task(monkeypatch, robot, 1)
    ref.seed(2)
    env._env.seed(2)
    assert np.array_equal(env._env.sample_spawn(), ref.sample_spawn())
