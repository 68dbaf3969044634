#!/usr/bin/env python3
"""State-only expert data collection for CubeStack-v0 (robot=franka) on the MI355X backend -- the caller's side of the hot path,
shaped like the reference's script (/root/reference/examples/franka/stack_cube_state.py:9-168).

`expert_policy` is the reference's (:18-95), constant for constant: per stage -- hover, grasp, lift, place, release (:139) -- a list of
(B, 9) joint targets: Cartesian waypoints from the hand's current position (`robot.get_link("hand").get_pos(envs_idx=...)`) to the
stage's target -- eight on a straight line, or, for place / release, four to a hover point 0.25 m above cube_2, four down to the
target and three repeats of the target -- one `robot.inverse_kinematics(..., init_qpos=previous solution)` per waypoint starting from
`robot.get_qpos(envs_idx=np.arange(B))`, then `80 // (waypoints - 1)` interpolated joint targets between consecutive waypoints (11 x 7 =
77 steps, 8 x 10 = 80 for place / release); targets: cube_1 + (0, 0, 0.25) / + (0, 0, 0.045) / + (0, 0, 0.28), cube_2 + (0, 0.004,
0.18) / + (0, 0, 0.18); fingers 0.04 open, -0.02 closed, closing over the last five targets of the grasp stage; quat (0, 1, 0, 0).
An env counts when ANY reward of the episode is positive (:156).  (The reference adds its offsets as CPU tensors to device tensors,
:35-52, which only runs on a CPU Genesis; here they live on the observation's device.)

`--policy tuned` keeps the schedule this repo shipped before (tools/stack_expert.py: 4 mm Cartesian steps, IK every step, grasp /
place heights tuned to this backend's finger geometry), which stacks more often on the re-stated Panda.

The reference writes a LeRobotDataset (lerobot is not installed here): the same features go to a compressed .npz.

    python examples/franka/stack_cube_state.py --num-envs 128 --out data/cube_stack.npz
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gym-genesis_amd"))

from gym_genesis.env import GenesisEnv  # noqa: E402

STAGES = ("hover", "grasp", "lift", "place", "release")  # stack_cube_state.py:139


def expert_policy(robot, obs, stage):
    """-> list of (B, 9) joint targets for `stage` (/root/reference/examples/franka/stack_cube_state.py:18-95)."""
    B, device = obs["agent_pos"].shape[0], obs["agent_pos"].device
    eef = robot.get_link("hand")
    quat_batch = torch.tensor([0, 1, 0, 0], dtype=torch.float32, device=device).unsqueeze(0).repeat(B, 1)  # (B, 4)
    cube1_pos = obs["environment_state"][:, :3]      # (B, 3)
    cube2_pos = obs["environment_state"][:, 11:14]   # (B, 3)
    grip_open, grip_closed = 0.04, -0.02
    v3 = lambda x, y, z: torch.tensor([x, y, z], device=device)  # noqa: E731
    hover_pos = None
    if stage == "hover":
        target_pos, grip_val = cube1_pos + v3(0.0, 0.0, 0.25), grip_open
    elif stage == "grasp":
        target_pos, grip_val = cube1_pos + v3(0.0, 0.0, 0.045), grip_closed
    elif stage == "lift":
        target_pos, grip_val = cube1_pos + v3(0.0, 0.0, 0.28), grip_closed
    elif stage == "place":
        hover_pos = cube2_pos + v3(0.0, 0.0, 0.25)
        target_pos, grip_val = cube2_pos + v3(0.0, 0.004, 0.18), grip_closed
    elif stage == "release":
        hover_pos = cube2_pos + v3(0.0, 0.0, 0.25)
        target_pos, grip_val = cube2_pos + v3(0.0, 0.0, 0.18), grip_open
    else:
        raise ValueError(f"Unknown stage: {stage}")
    # batched waypoints (:55-74)
    current_pos = robot.get_link("hand").get_pos(envs_idx=torch.arange(B, device=device))  # (B, 3)
    cart_wps = []
    if stage in ("place", "release"):
        for alpha in torch.linspace(0, 1, 4):   # current -> hover
            cart_wps.append((1 - alpha) * current_pos + alpha * hover_pos)
        for alpha in torch.linspace(0, 1, 4):   # hover -> target
            cart_wps.append((1 - alpha) * hover_pos + alpha * target_pos)
        for _ in range(3):                      # stabilise at the final position
            cart_wps.append(target_pos)
    else:
        for alpha in torch.linspace(0, 1, 8):   # current -> target
            cart_wps.append((1 - alpha) * current_pos + alpha * target_pos)
    # batched IK, each waypoint from the previous solution (:77-82)
    init_q = robot.get_qpos(envs_idx=np.arange(B))  # (B, 9)
    q_wps = []
    for wp in cart_wps:
        q = robot.inverse_kinematics(link=eef, pos=wp, quat=quat_batch, init_qpos=init_q)
        q_wps.append(q)
        init_q = q
    # joint targets between the waypoints (:85-91)
    num_interp = 80
    per = num_interp // (len(q_wps) - 1)
    path = []
    for i in range(len(q_wps) - 1):
        for t in range(per):
            alpha = t / (per - 1)
            path.append(((1 - alpha) * q_wps[i] + alpha * q_wps[i + 1]).clone())
    # fingers (:94-104)
    if stage == "grasp":
        for i in range(len(path) - 5):
            path[i][:, -2:] = grip_open
        for i in range(len(path) - 5, len(path)):
            alpha = (i - (len(path) - 5)) / 5
            path[i][:, -2:] = (1 - alpha) * grip_open + alpha * grip_closed
    else:
        for i in range(len(path)):
            path[i][:, -2:] = grip_val
    return path  # list of (B, 9)


def run_episode(env, obs, stages=STAGES, record=None):
    """One episode from the observation of a reset (:139-147) -> (states, env_states, actions, rewards), each (T, B, .) NumPy.
    `record`: a dict that receives the scene's spec, the state behind the reset and the actions (teacher-forced parity tests)."""
    if record is not None:
        mir = env._env._mir
        record["spec"] = mir.spec
        record["state0"] = [np.asarray(x.cpu()) for x in mir.get_state()]
        record["actions"] = []
    states, envs, acts, rews = [], [], [], []
    for stage in stages:
        for action in expert_policy(env.get_robot(), obs, stage):  # each action is (B, 9)
            obs, reward, done, _, _ = env.step(action)
            states.append(obs["agent_pos"]); envs.append(obs["environment_state"]); acts.append(action); rews.append(torch.as_tensor(reward))
            if record is not None:
                record["actions"].append(np.asarray(action.cpu()))
    return tuple(torch.stack([torch.as_tensor(t) for t in x]).cpu().numpy() for x in (states, envs, acts, rews))


def run_tuned(env, obs):
    """This repo's earlier schedule (tools/stack_expert.py): 4 mm Cartesian steps towards each stage's goal, IK every step."""
    B, dev = obs["agent_pos"].shape[0], obs["agent_pos"].device
    robot = env.get_robot()
    eef = robot.get_link("hand")
    quat = torch.tensor([0, 1, 0, 0], dtype=torch.float32, device=dev).expand(B, -1)
    c1, c2 = obs["environment_state"][:, :3].clone(), obs["environment_state"][:, 11:14].clone()
    z = lambda v: torch.tensor([0.0, 0.0, v], device=dev)  # noqa: E731
    OPEN, CLOSED, speed = 0.024, -0.01, 0.004
    stages = [(c1 + z(0.20), OPEN, 70), (c1 + z(0.058), OPEN, 60), (c1 + z(0.058), CLOSED, 30), (c1 + z(0.25), CLOSED, 70),
              (c2 + z(0.25), CLOSED, 90), (c2 + z(0.104), CLOSED, 70), (c2 + z(0.104), OPEN, 30), (c2 + z(0.25), OPEN, 50)]
    cur = obs["agent_pos"][:, :3].clone()
    states, envs, acts, rews = [], [], [], []
    for goal, grip, n in stages:
        for _ in range(n):
            d = goal - cur
            cur = cur + d * torch.clamp(speed / d.norm(dim=1, keepdim=True).clamp_min(1e-9), max=1.0)
            q = robot.inverse_kinematics(link=eef, pos=cur, quat=quat)
            action = torch.cat([q[:, :7], torch.full((B, 2), grip, device=dev)], 1)
            obs, reward, terminated, truncated, info = env.step(action)
            states.append(obs["agent_pos"]); envs.append(obs["environment_state"]); acts.append(action); rews.append(torch.as_tensor(reward))
    return tuple(torch.stack(x).cpu().numpy() for x in (states, envs, acts, rews))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=128)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--policy", choices=("reference", "tuned"), default="reference")
    ap.add_argument("--out", default=os.path.join("data", "cube_stack_state.npz"))
    args = ap.parse_args()
    B = args.num_envs
    env = GenesisEnv(task="cube_stack", robot="franka", num_envs=B)
    obs, _ = env.reset(seed=args.seed)
    states, envs, acts, rews = run_episode(env, obs) if args.policy == "reference" else run_tuned(env, obs)
    ok = np.where((rews > 0).any(axis=0))[0]  # any positive reward in the episode (:156)
    T = states.shape[0]
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    np.savez_compressed(args.out, fps=60, robot_type="franka",
                        **{"observation.state": np.concatenate([states[:, b] for b in ok]) if len(ok) else np.zeros((0, 9)),
                           "observation.environment_state": np.concatenate([envs[:, b] for b in ok]) if len(ok) else np.zeros((0, 14)),
                           "action": np.concatenate([acts[:, b] for b in ok]) if len(ok) else np.zeros((0, 9)),
                           "episode_index": np.repeat(np.arange(len(ok)), T), "frame_index": np.tile(np.arange(T), len(ok))})
    print(f"{len(ok)} / {B} envs had cube_1 stacked on cube_2 at some step ({T} steps, policy {args.policy}); wrote {len(ok) * T} frames to {args.out}")
    return len(ok)


if __name__ == "__main__":
    main()
