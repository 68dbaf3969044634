import sys, time, numpy as np, torch
sys.path.insert(0, "gym-genesis_amd")
import gym_genesis
for B, mode in ((1024, "global"), (4096, "global"), (1024, "per_env")):
    env = gym_genesis.make("gym_genesis/CubePick-v0", num_envs=B, enable_pixels=True, camera_capture_mode=mode)
    obs, info = env.reset(seed=0)
    e = env.unwrapped if hasattr(env, "unwrapped") else env
    a = np.stack([e.action_space.sample() for _ in range(B)])
    for _ in range(10): obs, r, term, trunc, info = env.step(a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): obs, r, term, trunc, info = env.step(a)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(100):
        obs, r, term, trunc, info = env.step(a); img = env.render()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"registry defaults (robot so101) B={B} {mode}: env.step {1e6*(t1-t0)/100:.0f} us | + render() {1e6*(t2-t1)/100:.0f} us | obs keys {list(obs)} pixels {tuple(obs['pixels'].shape)}")
