"""GPU probe: the reference's expert data-collection loop (IK every step + env.step) at 4096 envs: time per step split
into the batched IK call and the physics step."""
import os, sys, time
import torch
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
from gym_genesis.env import GenesisEnv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
obs, _ = env.reset(seed=0)
task = env._env; dev = task.device
robot, cube = env.get_robot(), obs["environment_state"][:, :3].clone()
eef = robot.get_link("hand")
quat = torch.tensor([0.0, 1.0, 0.0, 0.0], device=dev).repeat(B, 1)
stages = [("hover", 0.25, 0.04), ("stabilize", 0.25, 0.04), ("descend", 0.104, 0.04), ("grasp", 0.104, 0.0), ("lift", 0.40, 0.0)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for name, dz, grip in stages:
    tgt = cube + torch.tensor([0.0, 0.0, dz], device=dev)
    g = torch.full((B, 2), grip, device=dev)
    t_ik = t_st = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(40):
        ev[0].record()
        q = robot.inverse_kinematics(link=eef, pos=tgt, quat=quat)
        ev[1].record()
        a = torch.cat([q[:, :7], g], 1)
        task.step_raw(a.contiguous())
        ev[2].record()
        if k % 10 == 9:
            torch.cuda.synchronize(); t_ik += ev[0].elapsed_time(ev[1]); t_st += ev[1].elapsed_time(ev[2])
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 40
    print(f"{name:10s}: IK {t_ik / 4 * 1e3:6.1f} us  step(+cat) {t_st / 4 * 1e3:6.1f} us  wall {wall * 1e6:6.1f} us/step  -> {B / wall / 1e6:5.1f} M env-steps/s")
print("lifted", (task._reward == 1).float().mean().item())
