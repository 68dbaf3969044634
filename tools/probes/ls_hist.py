"""Line-search trips on the headline workload, from a -DMIR_AB_LSCOUNT build (niter | trips << 8 | two << 16 | tree0 << 20 | tree1 << 26)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(root, "gym-genesis_amd")]
import numpy as np, torch
from gym_genesis.env import GenesisEnv
B = 4096
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
env.reset(seed=0)
task = env._env
task._mir.set_diag(True)
gen = torch.Generator(device="cuda").manual_seed(1234)
acts = torch.empty((200, B, 9), dtype=torch.float32, device="cuda").uniform_(-1.0, 1.0, generator=gen)
tot = {k: 0 for k in ("waves", "iters", "trips", "two", "t0", "t1")}
trip_hist = np.zeros(64, np.int64)
for t in range(200):
    task.step_raw(acts[t])
    v = task._mir.get_diag()[2].cpu().numpy()
    w = v.reshape(-1, 4)
    niter, trips, two = w & 255, (w >> 8) & 255, (w >> 16) & 15
    t0, t1 = (w >> 20) & 63, (w >> 26) & 63
    tot["waves"] += w.shape[0]; tot["iters"] += int(niter.max(1).sum()); tot["trips"] += int(trips[:, 0].sum()); tot["two"] += int(two[:, 0].sum())
    tot["t0"] += int(t0.sum()); tot["t1"] += int(t1.sum())
    trip_hist += np.bincount(np.minimum(trips[:, 0], 63), minlength=64)
print(tot)
print("per wave-step: iterations %.3f, two-tree iterations %.3f, line-search trips %.3f; env-level unfinished tree-0 trips %.3f, tree-1 trips %.3f (per env-step)" % (
    tot["iters"] / tot["waves"], tot["two"] / tot["waves"], tot["trips"] / tot["waves"], tot["t0"] / tot["waves"] / 4, tot["t1"] / tot["waves"] / 4))
print("trips per wave-step histogram:", {i: int(h) for i, h in enumerate(trip_hist) if h})
