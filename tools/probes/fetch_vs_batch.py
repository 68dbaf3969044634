"""HBM-side reads of one step launch as a function of the batch: FETCH_SIZE(B) = fixed part (kernel code + model tables, fetched
once per XCD and launch) + per-env part.  Run under `rocprofv3 --pmc FETCH_SIZE` once per batch size (argv[1])."""
import os, sys
import torch
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
from gym_genesis.env import GenesisEnv
B = int(sys.argv[1])
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B)
env.reset(seed=0)
task = env._env
g = torch.Generator(device=task.device).manual_seed(1)
acts = torch.empty((16, B, 9), device=task.device).uniform_(-1, 1, generator=g)
for t in range(60): task.step_raw(acts[t % 16])
torch.cuda.synchronize()
