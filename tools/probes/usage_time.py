import sys, time, numpy as np, torch
sys.path.insert(0, "gym-genesis_amd")
from gym_genesis.env import GenesisEnv
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for task, robot, px in (("cube_pick", "franka", False), ("cube_pick", "franka", True), ("cube_stack", "franka", False), ("cube_stack", "so101", True), ("cube_pick", "so101", False)):
    B = 1024
    t0 = time.perf_counter()
    env = GenesisEnv(task=task, robot=robot, num_envs=B, enable_pixels=px)
    t1 = time.perf_counter()
    obs, info = env.reset(seed=0)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    a = np.stack([env.action_space.sample() for _ in range(B)])
    t3 = time.perf_counter()
    for _ in range(3): env.step(a)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    for _ in range(50): env.step(a)
    torch.cuda.synchronize(); t5 = time.perf_counter()
    img = env.render() if task == "cube_pick" else None  # (the batched stack tasks have no `cam`: env.render() raises there, as in the reference)
    torch.cuda.synchronize(); t6 = time.perf_counter()
    img = env.render() if task == "cube_pick" else None
    torch.cuda.synchronize(); t7 = time.perf_counter()
    print(f"{task} {robot} pixels={px}: create {1e3*(t1-t0):.1f} ms | first reset {1e3*(t2-t1):.1f} ms | action_space.sample() {1e3*(t3-t2):.2f} ms | first 3 steps {1e3*(t4-t3):.1f} ms | step {1e6*(t5-t4)/50:.0f} us | render() {1e3*(t6-t5):.2f} ms then {1e3*(t7-t6):.2f} ms, {type(img).__name__} {getattr(img, 'shape', None)}")
    env.close() if hasattr(env, "close") else None
