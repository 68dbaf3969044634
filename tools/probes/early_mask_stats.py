"""How many workgroups of a GenesisEnv.step launch send their terminated bytes from inside the solver loop (random-action workload)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]
import torch
from gym_genesis.env import GenesisEnv
B = 4096
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
env.reset(seed=0)
mir = env._env._mir
dev = env._env.device
g = torch.Generator(device=dev).manual_seed(0)
acts = [torch.empty((B, 9), device=dev).uniform_(-1, 1, generator=g) for _ in range(25)]
for t in range(100):
    env.step(acts[t % 25])
mir.set_diag(True)
mir.early_mask_stats(reset=True)
per = []
for t in range(100, 300):
    env.step(acts[t % 25])
    torch.cuda.synchronize()
    n, bad = mir.early_mask_stats(reset=True)
    per.append(n)
    assert bad == 0
import numpy as np
per = np.array(per)
print("workgroups (of 1024) that sent early per launch: mean %.1f min %d max %d; launches with all 1024 early: %d of %d" % (per.mean(), per.min(), per.max(), int((per == 1024).sum()), len(per)))
