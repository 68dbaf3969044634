"""Which part of the step kernel's HBM reads is write-allocate fill for the partial-line output stores?  Run under
`rocprofv3 --pmc FETCH_SIZE` (and again with WRITE_SIZE): 40 launches with all outputs (step_raw), 40 physics-only
(mir_step without output pointers), 40 with the packed row output (one 128-byte row per env)."""
import os, sys
import torch
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
from gym_genesis.env import GenesisEnv
B = 4096
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B)
env.reset(seed=0)
task = env._env; sc = task._mir
dev = task.device
g = torch.Generator(device=dev).manual_seed(1)
acts = torch.empty((64, B, 9), device=dev).uniform_(-1, 1, generator=g)
for t in range(10): task.step_raw(acts[t])
torch.cuda.synchronize()
for t in range(40): task.step_raw(acts[t % 64])
torch.cuda.synchronize()
for t in range(40): sc.step(1)
torch.cuda.synchronize()
rows = torch.zeros((B, 32), device=dev)
for t in range(40): sc.step_packed(acts[t % 64], rows)
torch.cuda.synchronize()
print("done")
