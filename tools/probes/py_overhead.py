import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gym-genesis_amd"))
import numpy as np, torch
from gym_genesis.tasks.fast_step import make_fast_step
B = 4096
class Mir:
    def __init__(s):
        s.buf = torch.empty(B * 21); s.term = torch.empty(B, dtype=torch.uint8)
    def step_go_ptr(s, p): pass
    def step_prepare_ptrs(s, p): pass
    def _alloc_outputs(s, a, e):
        buf = torch.empty(B * (a + e + 1)); term = torch.empty(B, dtype=torch.uint8); base = buf.data_ptr()
        outs = (buf[:a * B].view(B, a), buf[a * B:(a + e) * B].view(B, e), buf[(a + e) * B:], term)
        return outs, (base, base + 4 * a * B, base + 4 * (a + e) * B, term.data_ptr())
    def step_end_ptr(s, p): pass
    def as_action(s, a, d): return a
class Task: num_envs = B; device = torch.device("cpu")
step = make_fast_step(Task(), Mir(), 9, 9, 11)
acts = [torch.zeros((B, 9)) for _ in range(25)]
def loop(n):
    t = 0
    for _ in range(n):
        obs, r, term, trunc, info = step(acts[t % 25]); t += 1
        if term.any() or trunc.any() or t % 200 == 0: pass
for _ in range(3):
    t0 = time.perf_counter(); loop(20000); dt = (time.perf_counter() - t0) / 20000
    print(f"{dt*1e6:.2f} us per step (host only)")
def loop2(n):
    t = 0
    for _ in range(n):
        step(acts[t % 25]); t += 1
t0 = time.perf_counter(); loop2(20000); print(f"{(time.perf_counter()-t0)/20000*1e6:.2f} us per step, closure only")
