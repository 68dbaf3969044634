"""Scripted pick-and-stack on the device (GPU probe): the reference's expert loop shape (examples/franka/stack_cube_state.py:
hover -> grasp -> lift -> place -> release), targets from the batched IK every step, cube_1 stacked on cube_2."""
import os, sys
import numpy as np, torch
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
from gym_genesis.env import GenesisEnv


def run(B=64, seed=0, verbose=True, grasp_dz=0.062, place_dz=0.106, speed=0.004, record=None):
    """record: a dict that receives the scene spec, the state after reset (qpos, qvel, target, warmstart as NumPy) and the list of
    actions (for the teacher-forced state-parity test of tests/test_gpu_stack.py)"""
    env = GenesisEnv(task="cube_stack", robot="franka", num_envs=B)
    obs, _ = env.reset(seed=seed)
    if record is not None:
        record["spec"] = env._env._mir.spec
        record["state0"] = [x.cpu().numpy() for x in env._env._mir.get_state()]
        record["actions"] = []
    robot, dev = env.get_robot(), obs["agent_pos"].device
    eef = robot.get_link("hand")
    quat = torch.tensor([0, 1, 0, 0], dtype=torch.float32, device=dev).expand(B, -1)
    c1 = obs["environment_state"][:, :3].clone()
    c2 = obs["environment_state"][:, 11:14].clone()
    z = lambda v: torch.tensor([0.0, 0.0, v], device=dev)  # noqa: E731
    OPEN, CLOSED = 0.024, -0.01
    stages = [(c1 + z(0.20), OPEN, 70), (c1 + z(grasp_dz), OPEN, 60), (c1 + z(grasp_dz), CLOSED, 30), (c1 + z(0.25), CLOSED, 70),
              (c2 + z(0.25), CLOSED, 90), (c2 + z(place_dz), CLOSED, 70), (c2 + z(place_dz), OPEN, 30), (c2 + z(0.25), OPEN, 50)]
    cur = obs["agent_pos"][:, :3].clone()
    success = torch.zeros(B, dtype=torch.bool, device=dev)
    for goal, grip, n in stages:
        for _ in range(n):
            d = goal - cur
            dist = d.norm(dim=1, keepdim=True).clamp_min(1e-9)
            cur = cur + d * torch.clamp(speed / dist, max=1.0)
            q = robot.inverse_kinematics(link=eef, pos=cur, quat=quat)
            act = torch.cat([q[:, :7], torch.full((B, 2), grip, device=dev)], 1)
            if record is not None:
                record["actions"].append(act.cpu().numpy())
            obs, reward, term, trunc, info = env.step(act)
            success |= reward == 1
    final = (reward == 1)
    if verbose:
        es = obs["environment_state"]
        print(f"B={B}: stacked at the end {final.float().mean().item()*100:.0f} %, ever {success.float().mean().item()*100:.0f} %; "
              f"cube1 z {es[:, 2].mean().item():.3f}, cube2 z {es[:, 13].mean().item():.3f}")
    return final.float().mean().item(), success.float().mean().item()


if __name__ == "__main__":
    for g, p in ((0.062, 0.106), (0.058, 0.104), (0.066, 0.110)):
        print("grasp_dz", g, "place_dz", p, end=": ")
        run(grasp_dz=g, place_dz=p)
