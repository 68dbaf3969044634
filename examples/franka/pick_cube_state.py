#!/usr/bin/env python3
"""State-only expert data collection for CubePick-v0 on the MI355X backend -- the caller's side of the hot path, shaped like
the reference's script (/root/reference/examples/franka/pick_cube_state.py:14-120): a batched expert policy
(stage -> Cartesian target above the cube -> robot.inverse_kinematics -> joint targets + gripper), five stages of 40 steps
per episode, and the frames of every env that earned a reward kept as one episode.

The reference writes a LeRobotDataset (lerobot is not installed here); the same features -- "observation.state",
"observation.environment_state", "action", plus episode / frame indices -- are written to a compressed .npz instead.

The policy is the reference's, constant for constant (pick_cube_state.py:16-54,86-88 of the reference): stages hover, stabilize,
grasp, grasp, lift of 40 steps; hand target = the LIVE cube position + 0.115 / 0.115 / 0.03 / 0.03 / 0.25 m; finger targets 0.04 m
open, -0.02 m closed; quat (0, 1, 0, 0); IK from the current pose every step.  (The grasp target puts the hand origin 5 cm
above the floor, 6 cm lower than the fingertips allow: the fingers end up pressed on the floor around the cube, see
tests/test_ref_expert.py for what that exercises.)  `--stages tuned` selects the gentler schedule this repo used before
(hover 0.25, descend to 0.104 with open fingers, close to 0.0, lift to 0.40, targets relative to the cube's spawn position).

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

STAGES = ("hover", "stabilize", "grasp", "grasp", "lift")  # five stages of 40 steps (pick_cube_state.py:86)
TUNED_STAGES = ("hover", "stabilize", "descend", "close", "lift")


def expert_policy(robot, observation, stage):
    """(B, 9) joint-space action for `stage` -- the reference's expert_policy (pick_cube_state.py:14-54)."""
    agent_pos = observation["agent_pos"]                   # (B, 9)
    environment_state = observation["environment_state"]   # (B, 11)
    B, device = agent_pos.shape[0], agent_pos.device
    cube_pos = environment_state[:, :3]                    # the live cube position
    finder_pos = -0.02
    quat = torch.tensor([0, 1, 0, 0], dtype=torch.float32, device=device).expand(B, -1)
    eef = robot.get_link("hand")
    if stage in ("hover", "stabilize"):
        target_pos = cube_pos + torch.tensor([0.0, 0.0, 0.115], device=device)
        grip = torch.full((B, 2), 0.04, device=device)
    elif stage == "grasp":
        target_pos = cube_pos + torch.tensor([0.0, 0.0, 0.03], device=device)
        grip = torch.full((B, 2), finder_pos, device=device)
    elif stage == "lift":
        target_pos = cube_pos + torch.tensor([0.0, 0.0, 0.25], device=device)
        grip = torch.full((B, 2), finder_pos, device=device)
    else:
        raise ValueError(f"Unknown stage: {stage}")
    qpos = robot.inverse_kinematics(link=eef, pos=target_pos, quat=quat, envs_idx=torch.arange(B, device=device))  # (B, 9)
    return torch.cat([qpos[:, :-2], grip], dim=1)


def tuned_policy(robot, observation, stage, cube_ref):
    """This repo's earlier schedule (not the reference's): targets relative to the cube's spawn position `cube_ref`."""
    B, device = observation["agent_pos"].shape[0], observation["agent_pos"].device
    quat = torch.tensor([0, 1, 0, 0], dtype=torch.float32, device=device).expand(B, -1)
    eef = robot.get_link("hand")
    dz, grip = {"hover": (0.25, 0.04), "stabilize": (0.25, 0.04), "descend": (0.104, 0.04), "close": (0.104, 0.0), "lift": (0.40, 0.0)}[stage]
    target_pos = cube_ref + torch.tensor([0.0, 0.0, dz], device=device)
    qpos = robot.inverse_kinematics(link=eef, pos=target_pos, quat=quat, envs_idx=torch.arange(B, device=device))
    return torch.cat([qpos[:, :-2], torch.full((B, 2), grip, device=device)], dim=1)


def run_episode(env, obs, stages=STAGES, tuned=False):
    """One 5 x 40-step episode from the observation of a reset -> (states, env_states, actions, rewards), each (T, B, .) NumPy."""
    cube_ref = obs["environment_state"][:, :3].clone()
    states, envs, acts, rews = [], [], [], []
    for stage in stages:
        for _ in range(40):
            action = tuned_policy(env.get_robot(), obs, stage, cube_ref) if tuned else expert_policy(env.get_robot(), obs, stage)
            obs, reward, done, _, info = env.step(action)
            states.append(obs["agent_pos"]); envs.append(obs["environment_state"]); acts.append(action); rews.append(reward)
    return tuple(torch.stack([torch.as_tensor(t) for t in x]).cpu().numpy() for x in (states, envs, acts, rews))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=256)
    ap.add_argument("--episodes", type=int, default=2)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=os.path.join("data", "cube_pick_state.npz"))
    ap.add_argument("--stages", choices=("reference", "tuned"), default="reference")
    ap.add_argument("--exact-contacts", action="store_true",
                    help="keep every contact point (Genesis's behaviour): the envs whose narrowphase finds more than the 16-lane kernel's 16 -- "
                         "27 %% of this expert's env-steps, fingertips pressed on the floor around the cube -- are stepped by the wave-per-env kernel "
                         "instead of being thinned (DESIGN.md 5b)")
    args = ap.parse_args()

    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=args.num_envs, enable_pixels=False, exact_contacts=args.exact_contacts)
    env.reset(seed=args.seed)
    feats = {k: [] for k in ("observation.state", "observation.environment_state", "action", "episode_index", "frame_index")}
    kept = 0
    for ep in range(args.episodes):
        obs, _ = env.reset()
        tuned = args.stages == "tuned"
        states, envs, acts, rews = run_episode(env, obs, TUNED_STAGES if tuned else STAGES, tuned)  # (T, B, .)
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
