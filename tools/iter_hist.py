"""Histogram of Newton iterations per env-step on the headline workload (CubePick-v0, 4096 envs, U(-1,1) actions, 200 steps after a
reset) from the kernel's diagnostics.  Usage (GPU box): python3 tools/iter_hist.py [repo_root]  -- repo_root = another checkout to
compare against (tools/ab_repo.sh)."""
import os
import sys

root = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "gym-genesis_amd")]
import numpy as np
import torch

from gym_genesis.env import GenesisEnv

B = 4096
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
env.reset(seed=0)
task = env._env
task._mir.set_diag(True)
gen = torch.Generator(device="cuda").manual_seed(1234)
acts = torch.empty((200, B, 9), dtype=torch.float32, device="cuda").uniform_(-1.0, 1.0, generator=gen)
hist = np.zeros(64, np.int64)
ncon_hist = np.zeros(20, np.int64)
wave_max = []
for t in range(200):
    task.step_raw(acts[t])
    ncon, nefc, niter = (x.cpu().numpy() for x in task._mir.get_diag())
    hist += np.bincount(np.minimum(niter, 63), minlength=64)
    ncon_hist += np.bincount(np.minimum(ncon, 19), minlength=20)
    wave_max.append(niter.reshape(-1, 4).max(1))
wm = np.concatenate(wave_max)
print("root", root)
print("niter histogram (env-steps):", {i: int(h) for i, h in enumerate(hist) if h})
print("mean niter per env %.3f; mean of the wave maximum %.3f; 99th pct of wave maximum %d; max %d" % ((hist * np.arange(64)).sum() / hist.sum(), wm.mean(), np.percentile(wm, 99), wm.max()))
print("ncon histogram:", {i: int(h) for i, h in enumerate(ncon_hist) if h})
