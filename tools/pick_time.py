"""Step rate of the pick tasks on the 16-lane kernel for both robots (GPU probe): random PD targets around the home pose."""
import os, sys
import torch
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
from gym_genesis.env import GenesisEnv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for robot in ("franka", "so101"):
    env = GenesisEnv(task="cube_pick", robot=robot, num_envs=B, enable_pixels=False)
    env.reset(seed=0)
    task = env._env
    task._mir.set_diag(True)
    dev = task.device
    home = (task._home if hasattr(task, "_home") else task._zero)[0]
    gen = torch.Generator(device=dev).manual_seed(1)
    acts = home + 0.3 * torch.empty((256, B, home.numel()), device=dev).uniform_(-1, 1, generator=gen)
    for t in range(20): task.step_raw(acts[t])
    torch.cuda.synchronize()
    n = 300
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for t in range(n): task.step_raw(acts[t % 256])
    ev1.record(); torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) * 1e3 / n
    ncon, nefc, niter = (x.float().mean().item() for x in task._mir.get_diag())
    print(f"{robot} cube_pick B={B}: {us:.1f} us/step  {B/us:.2f} M env-steps/s  mean ncon {ncon:.1f} nefc {nefc:.1f} niter {niter:.2f}")
