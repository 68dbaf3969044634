"""The first env.step calls after a torch.cuda.synchronize(), as a function of how many steps ran before it: the launch call that follows
a synchronize pays for the runtime retiring the commands queued since the previous one (~0.4 us each)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]
import numpy as np, torch
from gym_genesis.env import GenesisEnv
B = 4096
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
env.reset(seed=0)
dev = env._env.device
g = torch.Generator(device=dev).manual_seed(0)
acts = [torch.empty((B, 9), device=dev).uniform_(-1, 1, generator=g) for _ in range(25)]
for _ in range(300):
    env.step(acts[0])
now = time.perf_counter_ns
for K in (10, 20, 40):
    R = 200
    res = np.zeros((R, 3))
    for r in range(R):
        for k in range(K):
            env.step(acts[k % 25])
        torch.cuda.synchronize(dev)
        t0 = now(); env.step(acts[0]); t1 = now(); env.step(acts[1]); t2 = now(); env.step(acts[2]); t3 = now()
        res[r] = [t1 - t0, t2 - t1, t3 - t2]
    m = np.median(res, 0) / 1e3
    print(f"{K} steps, synchronize, then three steps: {m[0]:.1f} {m[1]:.1f} {m[2]:.1f} us")
