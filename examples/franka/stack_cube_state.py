#!/usr/bin/env python3
"""State-only expert data collection for CubeStack-v0 (robot=franka) on the MI355X backend, after the reference's
/root/reference/examples/franka/stack_cube_state.py:14-140 (hover -> grasp -> lift -> place -> release with batched IK),
using the scripted policy of tools/stack_expert.py; frames of the envs that end with cube_1 stacked on cube_2 are written
to a compressed .npz with the LeRobot feature names.

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=128)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=os.path.join("data", "cube_stack_state.npz"))
    args = ap.parse_args()
    B = args.num_envs
    env = GenesisEnv(task="cube_stack", robot="franka", num_envs=B)
    obs, _ = env.reset(seed=args.seed)
    robot, dev = env.get_robot(), obs["agent_pos"].device
    eef = robot.get_link("hand")
    quat = torch.tensor([0, 1, 0, 0], dtype=torch.float32, device=dev).expand(B, -1)
    c1, c2 = obs["environment_state"][:, :3].clone(), obs["environment_state"][:, 11:14].clone()
    z = lambda v: torch.tensor([0.0, 0.0, v], device=dev)  # noqa: E731
    OPEN, CLOSED, speed = 0.024, -0.01, 0.004
    stages = [(c1 + z(0.20), OPEN, 70), (c1 + z(0.058), OPEN, 60), (c1 + z(0.058), CLOSED, 30), (c1 + z(0.25), CLOSED, 70),
              (c2 + z(0.25), CLOSED, 90), (c2 + z(0.104), CLOSED, 70), (c2 + z(0.104), OPEN, 30), (c2 + z(0.25), OPEN, 50)]
    cur = obs["agent_pos"][:, :3].clone()
    states, envs, acts = [], [], []
    for goal, grip, n in stages:
        for _ in range(n):
            d = goal - cur
            cur = cur + d * torch.clamp(speed / d.norm(dim=1, keepdim=True).clamp_min(1e-9), max=1.0)
            q = robot.inverse_kinematics(link=eef, pos=cur, quat=quat)
            action = torch.cat([q[:, :7], torch.full((B, 2), grip, device=dev)], 1)
            obs, reward, terminated, truncated, info = env.step(action)
            states.append(obs["agent_pos"]); envs.append(obs["environment_state"]); acts.append(action)
    ok = np.where(reward.cpu().numpy() == 1)[0]
    states, envs, acts = (torch.stack(x).cpu().numpy() for x in (states, envs, acts))
    T = states.shape[0]
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    np.savez_compressed(args.out, fps=60, robot_type="franka",
                        **{"observation.state": np.concatenate([states[:, b] for b in ok]) if len(ok) else np.zeros((0, 9)),
                           "observation.environment_state": np.concatenate([envs[:, b] for b in ok]) if len(ok) else np.zeros((0, 14)),
                           "action": np.concatenate([acts[:, b] for b in ok]) if len(ok) else np.zeros((0, 9)),
                           "episode_index": np.repeat(np.arange(len(ok)), T), "frame_index": np.tile(np.arange(T), len(ok))})
    print(f"{len(ok)} / {B} envs stacked cube_1 on cube_2; wrote {len(ok) * T} frames to {args.out}")
    return len(ok)


if __name__ == "__main__":
    main()
