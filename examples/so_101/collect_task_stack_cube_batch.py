#!/usr/bin/env python3
"""Expert data collection for CubeStack-v0 (robot=so101, batched) on the MI355X backend -- the caller's side of the hot path,
shaped like the reference's script (/root/reference/examples/so_101/collect_task_stack_cube_batch.py:12-230): per stage a batched
expert (`expert_policy_v2`, :24-116) returns a PATH of joint targets -- eight Cartesian waypoints from the gripper's current position
to the stage's target, one `robot.inverse_kinematics(..., init_qpos=previous waypoint's solution)` per waypoint, ten interpolated joint
targets between consecutive waypoints -- and the loop (:177-191) steps the env once per path entry.

The policy is the reference's, constant for constant: stages hover / grasp / lift / place / release / go_back in the loop (:173),
position_align / retreat defined beside them (:58-72); target offsets (0.01, 0.02, 0.25), (0.02, 0.02, 0.045), (0, 0, 0.28),
(0.006, 0, z_offset = 0.18) over cube_2, gripper_offset_z = -0.0981, correction_xy = (-0.005, 0.02); gripper 0.5 open / 0.1 closed, in the
grasp stage open until the last five path entries, then closing linearly (:104-109); go_back: ten targets from the current joint angles
to deg2rad(0, -177, 165, 72, -83, 0) with the gripper open (:72-81).  The orientation target is the reference's too, quirk included:
`R.from_euler('x', -90, degrees=True).as_quat()` is scipy's (x, y, z, w) = (-0.7071, 0, 0, 0.7071), handed to Genesis as (w, x, y, z)
(:32-33) -- i.e. a rotation of -90 degrees about z: reproduced literally.  What that constant MEANS, though, lives in the body frames of
the reference's MJCF, which is not in /root/reference (an un-vendored submodule): this repo's re-stated chain (backend/models.py) puts
the gripper link's x axis along the fingers, and no pose of it has the orientation (-0.7071, 0, 0, 0.7071).  `GRIPPER_FRAME` is the fixed
rotation between the two conventions, chosen so that the reference's target is what its top-down grasp evidently holds -- fingers
pointing straight down, q_ref * GRIPPER_FRAME = a rotation of +90 degrees about y here -- and is applied to the orientation target and
to nothing else.

The reference writes a LeRobotDataset with three videos per env (lerobot is not installed here; video recording is out of scope):
the same state / action features go to a compressed .npz for the envs whose LAST reward is positive (:207).  The SO-101 model of this
repo is re-stated from public specs (the reference's MJCF is an un-vendored submodule): what this script pins is the CALLER -- the
getters with `envs_idx=`, IK with `init_qpos` chaining, 360 contact-rich steps -- not a success rate.

    python examples/so_101/collect_task_stack_cube_batch.py --num-envs 128 --out data/so101_stack.npz
"""
import argparse
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gym-genesis_amd"))

from gym_genesis.env import GenesisEnv  # noqa: E402

STAGES = ("hover", "grasp", "lift", "place", "release", "go_back")  # collect_task_stack_cube_batch.py:173
# scipy's Rotation.from_euler('x', -90, degrees=True).as_quat() = (x, y, z, w), used as (w, x, y, z) by the reference (:32-33)
QUAT_AS_PASSED = (-math.sin(math.pi / 4), 0.0, 0.0, math.cos(math.pi / 4))
HOME_DEG = (0.0, -177.0, 165.0, 72.0, -83.0, 0.0)  # (:74)
# the gripper frame of the re-stated chain in the gripper frame the reference's constant is written in: Rz(+90 deg) * Ry(+90 deg), (w, x, y, z)
GRIPPER_FRAME = (0.5, -0.5, 0.5, 0.5)


def _qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return (aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw)


def expert_policy_v2(robot, obs, stage):
    """-> list of (B, 6) joint targets for `stage` (collect_task_stack_cube_batch.py:24-116)."""
    B, device = obs["agent_pos"].shape[0], obs["agent_pos"].device
    eef = robot.get_link("gripper")
    quat_batch = torch.tensor(_qmul(QUAT_AS_PASSED, GRIPPER_FRAME), dtype=torch.float32, device=device).repeat(B, 1)
    cube1_pos = obs["environment_state"][:, :3]      # (B, 3)
    cube2_pos = obs["environment_state"][:, 11:14]   # (B, 3)
    grip_open, grip_closed = 0.5, 0.1
    z_offset, gripper_offset_z = 0.18, -0.0981
    correction_xy = torch.tensor([-0.005, 0.02], device=device)
    v3 = lambda x, y, z: torch.tensor([x, y, z], device=device)  # noqa: E731
    if stage == "hover":
        target_pos, grip_val = cube1_pos + v3(0.01, 0.02, 0.25), grip_open
    elif stage == "grasp":
        target_pos, grip_val = cube1_pos + v3(0.02, 0.02, 0.045), grip_closed
    elif stage == "lift":
        target_pos, grip_val = cube1_pos + v3(0.0, 0.0, 0.28), grip_closed
    elif stage == "place":
        target_pos, grip_val = cube2_pos + v3(0.006, 0.0, z_offset), grip_closed
    elif stage == "position_align":
        target_pos = torch.cat([cube2_pos[:, :2] + correction_xy, (cube2_pos[:, 2] + z_offset - gripper_offset_z).unsqueeze(1)], dim=1)
        grip_val = grip_closed
    elif stage == "release":
        target_pos = torch.cat([cube2_pos[:, :2], (cube2_pos[:, 2] + z_offset - gripper_offset_z).unsqueeze(1)], dim=1)
        grip_val = grip_open
    elif stage == "retreat":
        target_pos = torch.cat([cube2_pos[:, :2] + correction_xy, (cube2_pos[:, 2] + 0.40 - gripper_offset_z).unsqueeze(1)], dim=1)
        grip_val = grip_open
    elif stage == "go_back":
        q_start = robot.get_qpos(envs_idx=np.arange(B))  # (B, 6)
        q_end = torch.deg2rad(torch.tensor(HOME_DEG, dtype=torch.float32, device=device)).repeat(B, 1)
        path = []
        for t in range(10):
            alpha = t / 9
            q = (1 - alpha) * q_start + alpha * q_end
            q[:, -1] = grip_open
            path.append(q.clone())
        return path
    else:
        raise ValueError(f"Unknown stage: {stage}")
    # eight Cartesian waypoints from where the gripper is (:86-87)
    current_pos = robot.get_link("gripper").get_pos(envs_idx=torch.arange(B))  # (B, 3)
    cart_wps = [(1 - alpha) * current_pos + alpha * target_pos for alpha in torch.linspace(0, 1, 8)]
    # one IK per waypoint, each started from the previous solution (:90-95)
    init_q = robot.get_qpos(envs_idx=np.arange(B))  # (B, 6)
    q_wps = []
    for wp in cart_wps:
        q = robot.inverse_kinematics(link=eef, pos=wp, quat=quat_batch, init_qpos=init_q)
        q_wps.append(q)
        init_q = q
    # ten joint targets per pair of waypoints (:98-103)
    path = []
    for i in range(len(q_wps) - 1):
        for t in range(10):
            alpha = t / 9
            path.append(((1 - alpha) * q_wps[i] + alpha * q_wps[i + 1]).clone())
    # gripper (:105-114)
    if stage == "grasp":
        for i in range(len(path) - 5):
            path[i][:, -1] = grip_open
        for i in range(len(path) - 5, len(path)):
            alpha = (i - (len(path) - 5)) / 5
            path[i][:, -1] = (1 - alpha) * grip_open + alpha * grip_closed
    else:
        for i in range(len(path)):
            path[i][:, -1] = grip_val
    return path


def run_episode(env, obs, stages=STAGES, record=None):
    """One episode from the observation of a reset -> (states, actions, rewards), each (T, B, .) NumPy (:177-191).  `record`: a dict that
    receives the scene's spec, the state behind the reset and the actions (for the teacher-forced parity tests)."""
    if record is not None:
        mir = env._env._mir
        record["spec"] = mir.spec
        record["state0"] = [np.asarray(x.cpu()) for x in mir.get_state()]
        record["actions"] = []
    states, acts, rews = [], [], []
    for stage in stages:
        for action in expert_policy_v2(env.get_robot(), obs, stage):  # (B, 6)
            obs, reward, done, _, _ = env.step(action)
            states.append(obs["agent_pos"]); acts.append(action); rews.append(torch.as_tensor(reward))
            if record is not None:
                record["actions"].append(np.asarray(action.cpu()))
    return tuple(torch.stack([torch.as_tensor(t) for t in x]).cpu().numpy() for x in (states, acts, rews))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=128)
    ap.add_argument("--episodes", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=os.path.join("data", "so101_stack_state.npz"))
    args = ap.parse_args()
    env = GenesisEnv(task="cube_stack", robot="so101", num_envs=args.num_envs, enable_pixels=False, strip_environment_state=False)
    env.reset(seed=args.seed)
    feats = {k: [] for k in ("observation.state", "action", "episode_index", "frame_index")}
    kept = 0
    for ep in range(args.episodes):
        obs, _ = env.reset()
        states, acts, rews = run_episode(env, obs)
        ok = np.where(rews[-1] > 0)[0]  # the envs whose final reward is positive (:207)
        T = states.shape[0]
        for b in ok:
            feats["observation.state"].append(states[:, b]); feats["action"].append(acts[:, b])
            feats["episode_index"].append(np.full(T, kept)); feats["frame_index"].append(np.arange(T))
            kept += 1
        print(f"episode {ep + 1}: {len(ok)} / {args.num_envs} envs end with cube_1 on cube_2 ({T} steps)")
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    np.savez_compressed(args.out, fps=30, robot_type="so101", **{k: np.concatenate(v) if v else np.zeros((0,)) for k, v in feats.items()})
    print(f"wrote {kept} successful episodes to {args.out}")
    return kept


if __name__ == "__main__":
    main()
