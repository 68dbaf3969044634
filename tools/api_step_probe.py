"""Where the time of one synchronous GenesisEnv.step goes (GPU box): per-piece host timings of the API path at B = 4096.
    python tools/api_step_probe.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gym_genesis.env import GenesisEnv  # noqa: E402

B, N = 4096, 3000
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
env.reset(seed=0)
task, mir = env._env, env._env._mir
dev = task.device
acts = list(torch.empty((64, B, 9), device=dev).uniform_(-1, 1).unbind(0))
pc = time.perf_counter


def avg(fn, n=N):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = pc()
    for i in range(n):
        fn()
    torch.cuda.synchronize()
    return (pc() - t0) / n * 1e6


i = [0]


def nxt():
    i[0] += 1
    return acts[i[0] & 63]


print(f"sync mode {mir.sync_mode}; null launch + completion round trip {mir.null_roundtrip_us(3000):.2f} us")
print(f"raw back-to-back launches      {avg(lambda: task.step_raw(nxt())):.2f} us/step")
print(f"env.step                       {avg(lambda: env.step(nxt())):.2f} us/step")


def begin_end():
    task.step_begin(nxt())
    task.step_end()


print(f"task.step_begin + step_end     {avg(begin_end):.2f} us/step")
bufs = (mir.empty(9), mir.empty(11), mir.empty(), mir.empty(dtype=torch.uint8))
ptrs = tuple(b.data_ptr() for b in bufs)
host = np.empty(B, np.bool_)
hp = host.ctypes.data


def c_only():
    mir.lib.mir_step_begin(mir.h, nxt().data_ptr(), ptrs[0], ptrs[1], ptrs[2], ptrs[3], mir._stream())
    mir.lib.mir_step_end(mir.h, hp)


print(f"ctypes begin + end only        {avg(c_only):.2f} us/step")
a0 = acts[0].data_ptr()
st = mir._stream()


def c_min():
    mir.lib.mir_step_begin(mir.h, a0, ptrs[0], ptrs[1], ptrs[2], ptrs[3], st)
    mir.lib.mir_step_end(mir.h, hp)


print(f"ctypes, all arguments cached   {avg(c_min):.2f} us/step")
# host-only costs (no GPU wait in them)
t0 = pc()
for _ in range(20000):
    mir._stream()
print(f"_stream()                      {(pc() - t0) / 20000 * 1e6:.2f} us")
t0 = pc()
for _ in range(20000):
    mir.as_action(acts[0], 9)
print(f"as_action()                    {(pc() - t0) / 20000 * 1e6:.2f} us")
t0 = pc()
for _ in range(20000):
    mir._alloc_outputs(9, 11)
print(f"_alloc_outputs()               {(pc() - t0) / 20000 * 1e6:.2f} us (overlapped with the kernel)")
t0 = pc()
for _ in range(20000):
    h2 = np.empty(B, np.bool_)
    h2.ctypes.data
print(f"np.empty + ctypes.data         {(pc() - t0) / 20000 * 1e6:.2f} us (overlapped)")
t0 = pc()
for _ in range(20000):
    host.any()
print(f"terminated.any()               {(pc() - t0) / 20000 * 1e6:.2f} us (the caller's loop)")
