"""Cost of env.reset() inside the README loop (reset-all every 200 steps in bench.py): wall time per reset and its parts."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "gym-genesis_amd"))
from gym_genesis.env import GenesisEnv
B = 4096
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B)
env.reset(seed=0)
task = env._env
def t(f, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print(f"env.reset()            {t(env.reset):8.1f} us")
print(f"task.reset()           {t(task.reset):8.1f} us")
print(f"sample_spawn()         {t(task.sample_spawn):8.1f} us")
pos_np = task.sample_spawn()
print(f"from_numpy().to(dev)   {t(lambda: torch.from_numpy(pos_np).to(task.device)):8.1f} us")
pos = torch.from_numpy(pos_np).to(task.device)
print(f"mir.reset              {t(lambda: task._mir.reset(pos, task._quat, task._home)):8.1f} us")
print(f"mir.step(1)            {t(lambda: task._mir.step(1)):8.1f} us")
print(f"get_obs                {t(task.get_obs):8.1f} us")
print(f"info list              {t(lambda: [False] * B):8.1f} us")
g = torch.Generator(device=task.device).manual_seed(1)
acts = torch.empty((64, B, 9), device=task.device).uniform_(-1, 1, generator=g)
def loop(n, every):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        env.step(acts[i % 64])
        if every and (i + 1) % every == 0: env.reset()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
loop(200, 0)
print(f"step loop, no resets   {loop(2000, 0):8.2f} us/step")
print(f"step loop, reset/200   {loop(2000, 200):8.2f} us/step")
