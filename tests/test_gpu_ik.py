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


def test_ik_rows_one_launch_equals_the_full_batch_launch_bit_for_bit():
    """robot.inverse_kinematics as the reference's experts call it (envs_idx = arange(B) with one quaternion expanded to the batch,
    /root/reference/examples/franka/pick_cube_state.py:46-51; init_qpos chained, envs_idx a tensor or a NumPy array,
    /root/reference/examples/so_101/collect_task_stack_cube_batch.py:90-95) is ONE launch of mir_inverse_kinematics_rows; the same
    arguments through the full-batch launch with torch scatter / gather around it (round 5's wrapper) give the same bits -- for every way
    the arguments can be addressed: by env, by row of a subset, a permutation of the batch, a broadcast quaternion, seeds for the arm's
    columns."""
    from gym_genesis.env import GenesisEnv
    from gym_genesis.tasks import views

    B = 256
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
    obs, _ = env.reset(seed=3)
    robot = env.get_robot()
    eef = robot.get_link("hand")
    dev = obs["agent_pos"].device
    g = torch.Generator(device="cpu").manual_seed(0)
    cube = obs["environment_state"][:, :3]
    pos_full = (cube + torch.tensor([0.0, 0.0, 0.2], device=dev)).contiguous()
    one = torch.tensor([0.0, 1.0, 0.0, 0.0], device=dev)
    quat_full = one.repeat(B, 1)
    quat_full[:, 1] += 0.1 * torch.rand(B, generator=g).to(dev)   # (not normalised: the kernel normalises)
    seed_full = robot.get_qpos() + 0.05 * torch.rand(B, 9, generator=g).to(dev)
    sub = torch.tensor([5, 0, 200, 17, 255, 3, 3], device=dev)     # (a repeated index: two rows of the same env)
    perm = torch.randperm(B, generator=g).to(dev)
    cases = [
        dict(pos=pos_full, quat=one.expand(B, -1), envs_idx=torch.arange(B, device=dev)),
        dict(pos=pos_full, quat=quat_full),
        dict(pos=pos_full, quat=None, init_qpos=seed_full),
        dict(pos=pos_full[sub], quat=quat_full[sub], envs_idx=sub),
        dict(pos=pos_full, quat=one, envs_idx=sub, init_qpos=seed_full[sub]),
        dict(pos=pos_full[sub], quat=quat_full, envs_idx=sub.cpu().numpy(), init_qpos=seed_full),
        dict(pos=pos_full, quat=quat_full, envs_idx=perm, init_qpos=seed_full),
        dict(pos=pos_full.cpu().numpy(), quat=[0.0, 1.0, 0.0, 0.0], envs_idx=list(range(B))),
    ]
    for i, kw in enumerate(cases):
        views.IK_ROWS = True
        q1, e1 = robot.inverse_kinematics(link=eef, return_error=True, **kw)
        views.IK_ROWS = False
        try:
            kw2 = dict(kw)
            if kw2.get("quat") is not None and torch.as_tensor(kw2["quat"]).numel() == 4:   # (the old wrapper wants a row per env / per index)
                n = B if kw2.get("envs_idx") is None else len(kw2["envs_idx"])
                kw2["quat"] = torch.as_tensor(kw2["quat"], dtype=torch.float32, device=dev).reshape(1, 4).repeat(n, 1)
            q0, e0 = robot.inverse_kinematics(link=eef, return_error=True, **kw2)
        finally:
            views.IK_ROWS = True
        assert q1.shape == q0.shape and torch.equal(q1, q0), f"case {i}: joint rows differ"
        assert torch.equal(e1, e0), f"case {i}: errors differ"
    with pytest.raises(ValueError):
        robot.inverse_kinematics(link=eef, pos=pos_full[:5], quat=one, envs_idx=sub)
