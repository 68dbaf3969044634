"""GPU probe: where the env.step() API path spends its wall time (allocation, launch, the mandated D->H sync)."""
import os, sys, time
import numpy as np, torch
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
from gym_genesis.env import GenesisEnv
B = 4096
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
env.reset(seed=0)
task = env._env; mir = task._mir; dev = task.device
gen = torch.Generator(device=dev).manual_seed(1)
acts = torch.empty((64, B, 9), device=dev).uniform_(-1, 1, generator=gen)
N = 300
def timeit(name, fn):
    for t in range(20): fn(t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(N): fn(t)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
    print(f"{name:58s} {dt*1e6:7.1f} us/step  {B/dt/1e6:7.1f} M env-steps/s")
timeit("env.step (current API path)", lambda t: env.step(acts[t % 64]))
def v_raw(t): task.step_raw(acts[t % 64])
timeit("step_raw (async hot path)", v_raw)
def v_sync(t):
    task.step_raw(acts[t % 64]); torch.cuda.current_stream().synchronize()
timeit("step_raw + stream.synchronize()", v_sync)
def v_cpu(t):
    task.step_raw(acts[t % 64]); task._term.cpu()
timeit("step_raw + term.cpu()", v_cpu)
pinned = torch.empty(B, dtype=torch.uint8).pin_memory()
def v_pin(t):
    task.step_raw(acts[t % 64]); pinned.copy_(task._term, non_blocking=True); torch.cuda.current_stream().synchronize()
timeit("step_raw + async copy to pinned + stream sync", v_pin)
ev = torch.cuda.Event()
def v_ev(t):
    task.step_raw(acts[t % 64]); ev.record(); ev.synchronize()
timeit("step_raw + event.synchronize()", v_ev)
def v_alloc(t):
    a = task._as_action(acts[t % 64])
    buf = torch.empty(B * 21, dtype=torch.float32, device=dev); term = torch.empty(B, dtype=torch.uint8, device=dev)
    mir.step_fused(a, buf[:9 * B].view(B, 9), buf[9 * B:20 * B].view(B, 11), buf[20 * B:], term)
    is_success = term.view(torch.bool)
    terminated = term.cpu().numpy().view(np.bool_)
    truncated = np.zeros(B, dtype=bool)
timeit("V1: 2 allocs, bool view, term.cpu()", v_alloc)
