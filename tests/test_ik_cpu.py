"""CPU-tier tests of the inverse-kinematics path: the oracle's damped-least-squares IK (the checker of
mir_inverse_kinematics) against forward kinematics, and the reference-facing surface
``robot.inverse_kinematics(link=eef, pos=..., quat=..., envs_idx=...)`` (examples/franka/pick_cube_state.py:46-51) on the
oracle-backed test double."""
import numpy as np
import pytest
import torch

import orc
from gym_genesis.backend import models

HOME = np.asarray(models.FRANKA_HOME, np.float64)


def _fk_hand(o, q9, e=0):
    q = o.read(orc.F_QPOS, e)
    q[:9] = q9
    o.write(orc.F_QPOS, q, e)
    o.fk(e)
    hand = o.spec.task.eef_body
    return o.read(orc.F_XPOS, e).reshape(-1, 3)[hand], o.read(orc.F_XQUAT, e).reshape(-1, 4)[hand]


def test_oracle_ik_reaches_reachable_poses_and_respects_limits():
    spec = models.franka_cube_pick_scene().build()
    B = 6
    o = orc.Oracle(spec, B)
    rng = np.random.default_rng(0)
    qt = np.tile(HOME, (B, 1))
    qt[:, :7] += rng.uniform(-0.5, 0.5, (B, 7))
    qt[:, 3] = np.clip(qt[:, 3], -2.9, -0.3)
    tp, tq = zip(*[_fk_hand(o, qt[e], e) for e in range(B)])
    q, err = o.ik(spec.task.eef_body, np.array(tp), np.array(tq), np.tile(HOME, (B, 1)), max_iters=100)
    assert (err[:, 0] < 5e-4).all() and (err[:, 1] < 5e-3).all()
    for e in range(B):
        p, qq = _fk_hand(o, q[e], e)
        assert np.abs(p - tp[e]).max() < 5e-4
        assert np.allclose(q[e, 7:], HOME[7:])  # fingers are not on the chain to the hand: untouched
    lo = np.array([spec.dof[i].range[0] for i in range(7)])
    hi = np.array([spec.dof[i].range[1] for i in range(7)])
    assert (q[:, :7] >= lo - 1e-9).all() and (q[:, :7] <= hi + 1e-9).all()
    # the pick pose of the reference's expert: hand pointing down (quat (0,1,0,0)) above a cube on the floor
    q2, err2 = o.ik(spec.task.eef_body, np.tile([[0.6, 0.1, 0.135]], (B, 1)), np.tile([[0, 1.0, 0, 0]], (B, 1)), np.tile(HOME, (B, 1)), max_iters=100)
    assert (err2[:, 0] < 5e-4).all() and (err2[:, 1] < 5e-3).all()
    # an unreachable target: the arm stretches towards it and stops at a finite error, inside the joint ranges
    q3, err3 = o.ik(spec.task.eef_body, np.tile([[3.0, 0.0, 0.5]], (B, 1)), None, np.tile(HOME, (B, 1)))
    assert (err3[:, 0] > 1.5).all() and np.isfinite(q3).all() and (q3[:, :7] >= lo - 1e-9).all() and (q3[:, :7] <= hi + 1e-9).all()
    # already there: zero iterations, the seed comes back bit-identical
    p0, q0 = _fk_hand(o, HOME, 0)
    q4, _ = o.ik(spec.task.eef_body, np.tile(p0, (B, 1)), np.tile(q0, (B, 1)), np.tile(HOME, (B, 1)))
    assert np.array_equal(q4, np.tile(HOME, (B, 1)))


def test_entity_inverse_kinematics_surface_on_the_test_double(monkeypatch):
    import fake_scene
    from gym_genesis.env import GenesisEnv
    from gym_genesis.tasks.franka import cube_pick

    monkeypatch.setattr(cube_pick, "MirScene", fake_scene.OracleScene)
    B = 3
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
    obs, _ = env.reset(seed=0)
    robot = env.get_robot()
    eef = robot.get_link("hand")
    cube = obs["environment_state"][:, :3]
    target = cube + torch.tensor([0.0, 0.0, 0.115])                       # "hover" stage of the expert (pick_cube_state.py:34-36)
    quat = torch.tensor([0, 1, 0, 0], dtype=torch.float32).expand(B, -1)
    qpos = robot.inverse_kinematics(link=eef, pos=target, quat=quat, envs_idx=torch.arange(B))
    assert qpos.shape == (B, 9) and qpos.dtype == torch.float32
    # drive the arm there with the env's own PD control: the hand ends up above the cube
    act = torch.cat([qpos[:, :-2], torch.full((B, 2), 0.04)], dim=1)
    for _ in range(60):
        obs, *_ = env.step(act)
    assert (obs["agent_pos"][:, :3] - target).abs().max() < 2e-2  # PD control droops ~1 cm under gravity at this reach
    sub = robot.inverse_kinematics(link=eef, pos=target[1:2], quat=quat[1:2], envs_idx=[1])
    assert sub.shape == (1, 9)
