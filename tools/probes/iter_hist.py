"""Newton iterations per env and step on the headline workload (solver diagnostics on): histogram over all envs, and the
histogram of the per-launch maximum -- the number the launch time follows."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]
import numpy as np, torch
from gym_genesis.env import GenesisEnv
B = 4096
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
env.reset(seed=0)
task = env._env; mir = task._mir
mir.set_diag(True)
g = torch.Generator(device=task.device).manual_seed(0)
allh = np.zeros(12, np.int64); maxh = np.zeros(12, np.int64); wgh = np.zeros(12, np.int64)
N = 400
for t in range(N):
    a = torch.empty((B, 9), device=task.device).uniform_(-1, 1, generator=g)
    env.step(a)
    if t % 200 == 199:
        env.reset()
    it = mir.get_diag()[2].cpu().numpy()
    allh += np.bincount(np.minimum(it, 11), minlength=12)
    maxh[min(int(it.max()), 11)] += 1
    wgh += np.bincount(np.minimum(it.reshape(-1, 4).max(1), 11), minlength=12)
print("iterations          :", list(range(8)))
print("envs (fraction)     :", np.round(allh[:8] / allh.sum(), 4))
print("workgroups (max of 4):", np.round(wgh[:8] / wgh.sum(), 4))
print("launches (max of all):", np.round(maxh[:8] / maxh.sum(), 3), " mean of max", round(float((maxh * np.arange(12)).sum() / maxh.sum()), 2))
