"""Strip-height sweep of the binned pixel kernel in bench.py's pixels state (1024 x 480 x 640, arms apart).  GPU box, repo root."""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
import torch
from gym_genesis.env import GenesisEnv
dev = torch.device("cuda", 0)
B, H, W = 1024, 480, 640
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=True, observation_height=H, observation_width=W, camera_capture_mode="per_env")
task = env._env
env.reset(seed=0)
gen = torch.Generator(device=dev).manual_seed(99)
for _ in range(20):
    task.step_raw(torch.empty((B, 9), dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=gen))
out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(4):
    for rows in (160, 224, 256, 320, 480):
        task._mir.debug_render_path(generic=False, strip_rows=rows)
        for _ in range(3): task.cam.render_envs(out=out)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(30): task.cam.render_envs(out=out)
        ev1.record(); torch.cuda.synchronize()
        print(rows, round(ev0.elapsed_time(ev1) * 1e3 / 30, 1), "us")
