"""Where one GenesisEnv.step() of the pick task goes on the host side: the launch call (go), the overlapped Python work, the wait for
the terminated bytes (end) and the Python between two calls.  Prints averages in us over 4000 steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]
import numpy as np, torch
from gym_genesis.env import GenesisEnv

B = 4096
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
env.reset(seed=0)
task = env._env
mir = task.mir if hasattr(task, "mir") else task._mir
g = torch.Generator(device=task.device).manual_seed(0)
acts = [torch.empty((B, 9), device=task.device).uniform_(-1, 1, generator=g) for _ in range(25)]
go, prepare, alloc, end = mir.step_go_ptr, mir.step_prepare_ptrs, mir._alloc_outputs, mir.step_end_ptr
now = time.perf_counter_ns
for _ in range(200):
    env.step(acts[0])
N = 4000
slot = alloc(9, 11); prepare(slot[1])


def spin(ns):
    t = now() + ns
    while now() < t:
        pass


# how much host time fits between two calls before the GPU starts to idle: a busy-wait of d us after end() returns
for d in (0, 1, 2, 3, 4, 6, 8):
    t_start = now()
    for t in range(N):
        go(acts[t % 25].data_ptr())
        host = np.empty(B, np.bool_)
        n = alloc(9, 11); prepare(n[1])
        end(host.ctypes.data)
        if d:
            spin(d * 1000)
    print(f"  extra {d} us between end() and the next launch: {(now() - t_start) / N / 1e3:.2f} us per step")
t_go = t_mid = t_end = t_between = 0
last = now()
for t in range(N):
    a = acts[t % 25].data_ptr()
    t0 = now(); go(a); t1 = now()
    host = np.empty(B, np.bool_)
    n = alloc(9, 11); prepare(n[1])
    t2 = now(); end(host.ctypes.data); t3 = now()
    slot = n
    t_between += t0 - last; t_go += t1 - t0; t_mid += t2 - t1; t_end += t3 - t2
    last = t3
tot = t_go + t_mid + t_end + t_between
print(f"split_step {mir.split_step}: go {t_go/N/1e3:.2f}  overlapped {t_mid/N/1e3:.2f}  end (wait) {t_end/N/1e3:.2f}  between {t_between/N/1e3:.2f}  total {tot/N/1e3:.2f} us")
