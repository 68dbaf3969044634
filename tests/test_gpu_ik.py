"""GPU tests of the batched inverse kinematics (mir_inverse_kinematics through the C ABI): parity with the oracle's
restatement of the same damped-least-squares iteration, the forward-kinematics property of the result, and the reference's
expert pick loop (examples/franka/pick_cube_state.py:27-56,86-93) run end to end on the device."""
import numpy as np
import pytest
import torch

import orc
from gym_genesis.backend import models

pytestmark = pytest.mark.gpu

HOME = np.asarray(models.FRANKA_HOME, np.float32)


def test_ik_matches_oracle_and_forward_kinematics():
    from gym_genesis.backend.lib import MirScene

    spec = models.franka_cube_pick_scene().build()
    B = 64
    sc = MirScene(spec, B)
    o = orc.Oracle(spec, B)
    rng = np.random.default_rng(3)
    # reachable targets: the hand pose at random joint configurations
    qt = np.tile(HOME, (B, 1)).astype(np.float64)
    qt[:, :7] += rng.uniform(-0.6, 0.6, (B, 7))
    qt[:, 3] = np.clip(qt[:, 3], -2.9, -0.3)
    q16 = np.zeros((B, 16), np.float32)
    q16[:, :9] = qt
    q16[:, 9:12] = [0.6, 0.0, 0.02]
    q16[:, 12] = 1.0
    sc.set_state(qpos=q16)
    pos, quat = (t.cpu().numpy() for t in sc.get_links())
    hand = spec.task.eef_body
    tp, tq = pos[:, hand], quat[:, hand]
    seed = np.tile(HOME, (B, 1))
    q, err = sc.inverse_kinematics(hand, tp, tq, seed, return_error=True, max_iters=100)
    q, err = q.cpu().numpy(), err.cpu().numpy()
    qo, erro = o.ik(hand, tp, tq, seed, max_iters=100)
    assert (err[:, 0] < 5e-4).all() and (err[:, 1] < 5e-3).all() and (erro[:, 0] < 5e-4).all()
    # same iteration, float32 vs float64: the iterates agree until the tolerance test fires
    assert np.abs(q - qo).max() < 5e-3, np.abs(q - qo).max()
    assert np.median(np.abs(q - qo).max(1)) < 1e-4
    # property: forward kinematics of the solution is the target
    q16[:, :9] = q
    sc.set_state(qpos=q16)
    pos2, quat2 = (t.cpu().numpy() for t in sc.get_links())
    assert np.abs(pos2[:, hand] - tp).max() < 6e-4
    assert np.allclose(q[:, 7:], HOME[7:])  # finger joints are off the chain
    # position-only target and the default seed (= current scene state)
    q3 = sc.inverse_kinematics(hand, tp, None).cpu().numpy()
    assert np.abs(q3 - q).max() < 1e-3  # already at a solution: (almost) no motion


def test_expert_pick_policy_end_to_end_on_device():
    """hover -> stabilize -> descend -> grasp -> lift, 40 steps each, targets from batched IK every step, as the reference's
    data-collection loop does (pick_cube_state.py:86-88: the hover target is held for two stages, which the wrist joints --
    +-12 N m -- need to settle); heights adapted to this repo's box-pad fingers (tests/golden/make_grasp_targets.py)."""
    from gym_genesis.env import GenesisEnv

    B = 64
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
    obs, _ = env.reset(seed=0)
    robot = env.get_robot()
    eef = robot.get_link("hand")
    dev = obs["agent_pos"].device
    quat = torch.tensor([0, 1, 0, 0], dtype=torch.float32, device=dev).expand(B, -1)
    cube0 = obs["environment_state"][:, :3].clone()
    success = torch.zeros(B, dtype=torch.bool, device=dev)
    for stage, dz, grip in (("hover", 0.25, 0.04), ("stabilize", 0.25, 0.04), ("descend", 0.104, 0.04), ("grasp", 0.104, 0.0), ("lift", 0.40, 0.0)):
        for _ in range(40):
            target = cube0 + torch.tensor([0.0, 0.0, dz], device=dev)
            qpos = robot.inverse_kinematics(link=eef, pos=target, quat=quat, envs_idx=torch.arange(B, device=dev))
            action = torch.cat([qpos[:, :-2], torch.full((B, 2), grip, device=dev)], dim=1)
            obs, reward, terminated, truncated, info = env.step(action)
            success |= reward == 1
    frac = success.float().mean().item()
    assert frac > 0.8, f"only {frac:.2f} of the envs lifted the cube"
    print(f"expert pick with on-device IK: {frac * 100:.0f} % of {B} envs lifted the cube")


def test_expert_pick_and_stack_end_to_end_on_device():
    """The stack task's whole loop on the device: hover -> grasp -> lift -> place -> release (the stage list of
    examples/franka/stack_cube_state.py:33-55), Cartesian targets through the batched IK every step, the wave-per-env kernel
    carrying finger-cube and cube-cube contacts, the stack reward |dxy| < 0.05 and dz > 0.03 (cube_stack_kitchen_batch.py:138-146)."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stack_expert

    final, ever = stack_expert.run(B=32, seed=1, verbose=False, grasp_dz=0.058, place_dz=0.104)
    assert final > 0.8, f"only {final:.2f} of the envs ended with cube_1 stacked on cube_2"
    print(f"expert pick-and-stack with on-device IK: {final * 100:.0f} % stacked at the end ({ever * 100:.0f} % at some point)")
