#!/usr/bin/env python3
"""State-only expert data collection for CubePick-v0 on the MI355X backend -- the caller's side of the hot path, shaped like
the reference's script (/root/reference/examples/franka/pick_cube_state.py:14-120): a batched expert policy
(stage -> Cartesian target above the cube -> robot.inverse_kinematics -> joint targets + gripper), five stages of 40 steps
per episode, and the frames of every env that earned a reward kept as one episode.

The reference writes a LeRobotDataset (lerobot is not installed here); the same features -- "observation.state",
"observation.environment_state", "action", plus episode / frame indices -- are written to a compressed .npz instead.
Stage heights are the ones that suit this repo's box-pad fingers (see tests/golden/make_grasp_targets.py).

    python examples/franka/pick_cube_state.py --num-envs 256 --episodes 2 --out data/cube_pick.npz
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gym-genesis_amd"))

from gym_genesis.env import GenesisEnv  # noqa: E402

STAGES = ("hover", "stabilize", "descend", "grasp", "lift")  # five stages of 40 steps (pick_cube_state.py:86)


def expert_policy(robot, observation, stage, cube_ref):
    """(B, 9) joint-space action for `stage` (pick_cube_state.py:14-56)."""
    B, device = observation["agent_pos"].shape[0], observation["agent_pos"].device
    quat = torch.tensor([0, 1, 0, 0], dtype=torch.float32, device=device).expand(B, -1)  # hand pointing down
    eef = robot.get_link("hand")
    dz, grip = {"hover": (0.25, 0.04), "stabilize": (0.25, 0.04), "descend": (0.104, 0.04), "grasp": (0.104, 0.0), "lift": (0.40, 0.0)}[stage]
    target_pos = cube_ref + torch.tensor([0.0, 0.0, dz], device=device)
    qpos = robot.inverse_kinematics(link=eef, pos=target_pos, quat=quat, envs_idx=torch.arange(B, device=device))  # (B, 9)
    return torch.cat([qpos[:, :-2], torch.full((B, 2), grip, device=device)], dim=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=256)
    ap.add_argument("--episodes", type=int, default=2)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=os.path.join("data", "cube_pick_state.npz"))
    args = ap.parse_args()

    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=args.num_envs, enable_pixels=False)
    env.reset(seed=args.seed)
    feats = {k: [] for k in ("observation.state", "observation.environment_state", "action", "episode_index", "frame_index")}
    kept = 0
    for ep in range(args.episodes):
        obs, _ = env.reset()
        cube_ref = obs["environment_state"][:, :3].clone()
        states, envs, acts, rews = [], [], [], []
        for stage in STAGES:
            for _ in range(40):
                action = expert_policy(env.get_robot(), obs, stage, cube_ref)
                obs, reward, done, _, info = env.step(action)
                states.append(obs["agent_pos"]); envs.append(obs["environment_state"]); acts.append(action); rews.append(reward)
        states, envs, acts, rews = (torch.stack(x).cpu().numpy() for x in (states, envs, acts, rews))  # (T, B, .)
        ok = np.where((rews > 0).any(axis=0))[0]  # keep the envs that earned a reward (pick_cube_state.py:107-118)
        for b in ok:
            T = states.shape[0]
            feats["observation.state"].append(states[:, b]); feats["observation.environment_state"].append(envs[:, b])
            feats["action"].append(acts[:, b]); feats["episode_index"].append(np.full(T, kept)); feats["frame_index"].append(np.arange(T))
            kept += 1
        print(f"episode {ep + 1}: {len(ok)} / {args.num_envs} envs lifted the cube")
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    np.savez_compressed(args.out, fps=60, robot_type="franka", **{k: np.concatenate(v) if v else np.zeros((0,)) for k, v in feats.items()})
    print(f"wrote {kept} successful episodes ({kept * 200} frames) to {args.out}")
    return kept


if __name__ == "__main__":
    main()
