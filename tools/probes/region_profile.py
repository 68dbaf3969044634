"""Where a 20-step timed region of bench.py goes: host time of each of its env.step calls after a synchronize, and the closing synchronize."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]
import numpy as np, torch
from gym_genesis.env import GenesisEnv
B, K, R = 4096, 20, 300
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
env.reset(seed=0)
dev = env._env.device
g = torch.Generator(device=dev).manual_seed(0)
acts = [torch.empty((B, 9), device=dev).uniform_(-1, 1, generator=g) for _ in range(25)]
for t in range(200):
    env.step(acts[t % 25])
now = time.perf_counter_ns
step_t = np.zeros((R, K)); sync_t = np.zeros(R); tot = np.zeros(R)
for r in range(R):
    torch.cuda.synchronize(dev)
    t0 = now(); last = t0
    for k in range(K):
        obs, rew, term, trunc, info = env.step(acts[(r + k) % 25])
        if term.any() or trunc.any():
            pass
        t = now(); step_t[r, k] = t - last; last = t
    torch.cuda.synchronize(dev)
    t1 = now(); sync_t[r] = t1 - last; tot[r] = t1 - t0
print("host time of step k after a synchronize (us, median over regions):", np.round(np.median(step_t, 0) / 1e3, 1))
print("closing synchronize: median %.1f us; whole region median %.1f us = %.2f us per step" % (np.median(sync_t) / 1e3, np.median(tot) / 1e3, np.median(tot) / 1e3 / K))
