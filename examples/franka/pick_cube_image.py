#!/usr/bin/env python3
"""Image expert data collection for CubePick-v0 on the MI355X backend: the caller's side of the `pixels` path, shaped like the
reference's script (/root/reference/examples/franka/pick_cube_image.py:9-15,70-117): the same batched expert as pick_cube_state.py
(imported from there), `enable_pixels=True, camera_capture_mode="per_env", strip_environment_state=False`, five stages of 40 steps,
and for every env that earned a reward one episode of ("observation.state", "action", "observation.image").

The reference stacks the whole batch's images on the HOST -- its own FIXME: "system ram crash if B is too big" -- and hands them to
a LeRobotDataset with `use_videos=True`.  Here the (T, B, H, W, 3) frames stay in HBM (200 x 64 envs x 480 x 640 x 3 = 11.8 GB of
288); only the envs that lifted the cube are brought to the host, and their 200 frames are written as one Motion-JPEG `.mp4` per
episode (gym_genesis/tasks/video.py) beside an `.npz` with the LeRobot feature names.

    python examples/franka/pick_cube_image.py --num-envs 16 --episodes 1 --out data/cube_pick_image
"""
import argparse
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "gym-genesis_amd"))

from gym_genesis.env import GenesisEnv  # noqa: E402
from gym_genesis.tasks.video import write_mjpeg_mp4  # noqa: E402

_spec = importlib.util.spec_from_file_location("pick_cube_state", os.path.join(HERE, "pick_cube_state.py"))
state_script = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(state_script)


def run_episode(env, obs, tuned=False):
    """One 5 x 40-step episode -> states (T, B, 9), actions (T, B, 9), rewards (T, B) as NumPy, images (T, B, H, W, 3) uint8 ON THE DEVICE."""
    stages = state_script.TUNED_STAGES if tuned else state_script.STAGES
    cube_ref = obs["environment_state"][:, :3].clone()
    T, (B, H, W, _) = 40 * len(stages), obs["pixels"].shape
    images = torch.empty((T, B, H, W, 3), dtype=torch.uint8, device=obs["pixels"].device)
    states, acts, rews = [], [], []
    robot = env.get_robot()
    for stage in stages:
        for _ in range(40):
            action = state_script.tuned_policy(robot, obs, stage, cube_ref) if tuned else state_script.expert_policy(robot, obs, stage)
            obs, reward, done, _, info = env.step(action)
            images[len(states)].copy_(obs["pixels"])          # (the observation's image buffer is reused by later steps)
            states.append(obs["agent_pos"]); acts.append(action); rews.append(reward)
    return tuple(torch.stack([torch.as_tensor(t) for t in x]).cpu().numpy() for x in (states, acts, rews)) + (images,)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=16)
    ap.add_argument("--episodes", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--out", default=os.path.join("data", "cube_pick_image"), help="directory: episodes.npz + videos/episode_XXXXXX.mp4")
    ap.add_argument("--stages", choices=("reference", "tuned"), default="reference")
    args = ap.parse_args()

    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=args.num_envs, enable_pixels=True, observation_height=args.height,
                     observation_width=args.width, camera_capture_mode="per_env", strip_environment_state=False)
    env.reset(seed=args.seed)
    os.makedirs(os.path.join(args.out, "videos"), exist_ok=True)
    feats = {k: [] for k in ("observation.state", "action", "episode_index", "frame_index")}
    kept = 0
    for ep in range(args.episodes):
        obs, _ = env.reset()
        states, acts, rews, images = run_episode(env, obs, args.stages == "tuned")
        ok = np.where((rews > 0).any(axis=0))[0]  # keep the envs that earned a reward (pick_cube_image.py:101-114)
        for b in ok:
            T = states.shape[0]
            write_mjpeg_mp4(os.path.join(args.out, "videos", f"episode_{kept:06d}.mp4"), list(images[:, b].cpu().numpy()), fps=60)
            feats["observation.state"].append(states[:, b]); feats["action"].append(acts[:, b])
            feats["episode_index"].append(np.full(T, kept)); feats["frame_index"].append(np.arange(T))
            kept += 1
        print(f"episode {ep + 1}: {len(ok)} / {args.num_envs} envs lifted the cube")
    np.savez_compressed(os.path.join(args.out, "episodes.npz"), fps=60, robot_type="franka", task="pick cube",
                        **{k: np.concatenate(v) if v else np.zeros((0,)) for k, v in feats.items()})
    print(f"wrote {kept} successful episodes ({kept * 200} frames) to {args.out}")
    return kept


if __name__ == "__main__":
    main()
