"""Per-stage step time of the scripted pick at 4096 envs (GPU probe): which phases of the manipulation are expensive."""
import os, sys
import torch
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
from gym_genesis.env import GenesisEnv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
obs, _ = env.reset(seed=0)
task = env._env; dev = task.device
task._mir.set_diag(True)
robot, cube = env.get_robot(), obs["environment_state"][:, :3].clone()
eef = robot.get_link("hand")
quat = torch.tensor([0.0, 1.0, 0.0, 0.0], device=dev).repeat(B, 1)
stages = [("hover", 0.25, 0.04), ("stabilize", 0.25, 0.04), ("descend", 0.104, 0.04), ("grasp", 0.104, 0.0), ("lift", 0.40, 0.0)]
q_prev = None
for name, dz, grip in stages:
    q = robot.inverse_kinematics(link=eef, pos=cube + torch.tensor([0.0, 0.0, dz], device=dev), quat=quat, init_qpos=q_prev)
    q_prev = q
    tg = torch.cat([q[:, :7], torch.full((B, 2), grip, device=dev)], 1).contiguous()
    for half in range(2):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(20): task.step_raw(tg)
        ev1.record(); torch.cuda.synchronize()
        ncon, nefc, niter = task._mir.get_diag()
        print(f"{name:10s} steps {20*half:2d}-{20*half+19:2d}: {ev0.elapsed_time(ev1)*1e3/20:6.1f} us/step  ncon mean {ncon.float().mean():.1f} max {ncon.max().item()}  niter mean {niter.float().mean():.2f} max {niter.max().item()}")
print("lifted", (task._reward == 1).float().mean().item())
