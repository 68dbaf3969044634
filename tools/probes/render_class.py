"""Can the rasteriser choose its output buffer's placement class (the same kernel takes 175-183 us into some 0.94 GB
allocations and 204-216 us into others)?  Candidates are allocated one after the other and HELD, the per-env render of bench.py's
pixel workload is timed into each (10 warm-up + 30 timed renders), and the question is how many candidates a product would have to
try before it holds a fast one.  Then everything is freed and three more are allocated: do they land where the first ones did?
Usage: python3 tools/probes/render_class.py [candidates]"""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
import numpy as np, torch
from gym_genesis.env import GenesisEnv
dev = torch.device("cuda", 0)
B, H, W = 1024, 480, 640
NC = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ALLOC_MB = int(sys.argv[2]) if len(sys.argv) > 2 else 0     # allocate this many MB per candidate (0: exactly the image bytes, 900 MB) and draw into its head
CHURN = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # allocate and free this many GB in odd sizes first (a process that has lived a while)
env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=True, observation_height=H, observation_width=W, camera_capture_mode="per_env")
task = env._env
env.reset(seed=0)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(out, n=30):
    for _ in range(10): task.cam.render_envs(out=out)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(n): task.cam.render_envs(out=out)
    ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) * 1e3 / n
warm = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev)
for _ in range(200): task.cam.render_envs(out=warm)
print("warm buffer %x: %.1f us" % (warm.data_ptr(), t(warm)))
if CHURN:
    junk = [torch.empty(int((37 + 91 * (i % 7)) * 2**20), dtype=torch.uint8, device=dev) for i in range(CHURN * 3)]
    del junk[::2]
    torch.cuda.empty_cache()
    junk2 = [torch.empty(int(513 * 2**20), dtype=torch.uint8, device=dev) for i in range(CHURN)]
    del junk, junk2
    torch.cuda.empty_cache()
bufs, times = [], []
NB = B * H * W * 3
for k in range(NC):
    if ALLOC_MB:
        raw = torch.empty(ALLOC_MB * 2**20, dtype=torch.uint8, device=dev)
        b = raw[:NB].view(B, H, W, 3)
    else:
        b = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev)
    bufs.append(b); times.append(t(b))
times = np.array(times)
best = times.min()
fast = times <= 1.04 * best
print("candidates, us:", " ".join("%.0f%s" % (x, "*" if f else "") for x, f in zip(times, fast)))
print("addresses GB:", " ".join("%.2f" % (b.data_ptr() / 2**30) for b in bufs))
first_fast = int(np.argmax(fast))
print(f"best {best:.1f} us, worst {times.max():.1f} us; fast class (within 4 % of the best) in {int(fast.sum())} of {NC}; first fast candidate: #{first_fast} "
      f"(a product probing up to 3 candidates would hold a fast one: {bool(fast[:3].any())})")
addr0 = [b.data_ptr() for b in bufs[:3]]
del bufs, b
torch.cuda.empty_cache()
print(f"(alloc {ALLOC_MB or 900} MB per candidate, churn {CHURN} GB)")
again = [torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(3)]
print("after freeing everything, three new allocations: addresses", ["%.2f" % (x.data_ptr() / 2**30) for x in again], "same as the first three:", [x.data_ptr() for x in again] == addr0,
      "us:", ["%.0f" % t(x) for x in again])
print("warm buffer again: %.1f us" % t(warm))
