"""Time of the GLOBAL view (camera_capture_mode="global", the registry's default: one H x W image of all envs at their grid offsets;
also GenesisEnv.render()).  GPU box, repo root."""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
import torch
from gym_genesis.env import GenesisEnv
for B in (64, 1024, 4096):
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=True, camera_capture_mode="global")
    env.reset(seed=0)
    task = env._env
    for _ in range(5): img = task.cam.render_global()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20): img = task.cam.render_global()
    ev1.record(); torch.cuda.synchronize()
    print(f"B={B}: global view {tuple(img.shape)}: {ev0.elapsed_time(ev1) / 20 * 1e3:.1f} us per render")
    del env, task
