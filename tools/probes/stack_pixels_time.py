import sys, time, torch, numpy as np
sys.path.insert(0, "gym-genesis_amd")
from gym_genesis.env import GenesisEnv
for robot in ("franka", "so101"):
    for mode in ("global", "per_env"):
        B = 256
        env = GenesisEnv(task="cube_stack", robot=robot, num_envs=B, enable_pixels=True, observation_height=96, observation_width=128, camera_capture_mode=mode)
        obs, _ = env.reset(seed=0)
        shapes = {k: tuple(v.shape) for k, v in obs["pixels"].items()}
        act = torch.as_tensor(np.tile(np.asarray(env._env._home_qpos(), np.float32), (B, 1)), device=env._env.device)
        for _ in range(5): env.step(act)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): env.step(act)
        torch.cuda.synchronize()
        print(robot, mode, shapes, f"{(time.perf_counter() - t0) / 30 * 1e6:.0f} us per env.step with pixels (B={B})")
        del env
