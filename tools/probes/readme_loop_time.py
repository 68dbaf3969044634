import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gym-genesis_amd"))
from gym_genesis.env import GenesisEnv
for B in (1024, 4096):
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=True, camera_capture_mode="global")
    env.reset(seed=0)
    a = np.random.default_rng(0).uniform(-1, 1, (B, 9)).astype(np.float32)
    for _ in range(20): env.step(a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): obs, r, term, trunc, info = env.step(a)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(200):
        obs, r, term, trunc, info = env.step(a); img = env.render()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    frames = []
    for _ in range(200):
        obs, r, term, trunc, info = env.step(a); frames.append(env.render())   # (a caller that keeps the frames: every array is new memory)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"B={B}: the README loop with the frames kept {1e6*(t3-t2)/200:.0f} us per iteration")
    print(f"B={B} global pixels: env.step {1e6*(t1-t0)/200:.0f} us | env.step + env.render() (the README loop) {1e6*(t2-t1)/200:.0f} us; pixels {tuple(obs['pixels'].shape)} {obs['pixels'].device}")
