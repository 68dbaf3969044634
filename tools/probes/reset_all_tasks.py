"""env.reset() of every task at 4096 envs: wall time per call (host RNG mirror of the reference + the reset launch).  GPU box, repo root."""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
import torch
from gym_genesis.env import GenesisEnv
B = 4096
for task, robot in (("cube_pick", "franka"), ("cube_pick", "so101"), ("cube_stack", "franka"), ("cube_stack", "so101")):
    env = GenesisEnv(task=task, robot=robot, num_envs=B)
    env.reset(seed=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): env.reset()
    torch.cuda.synchronize()
    print(f"{task} {robot}: reset {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
    del env
