"""Does the pixel kernel's time depend on where the output buffer sits?  bench.py's pixels state, one pool, views at several offsets,
and several separately allocated buffers.  GPU box, repo root."""
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
N = B * H * W * 3
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(out, n=60):
    for _ in range(20): task.cam.render_envs(out=out)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(n): task.cam.render_envs(out=out)
    ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) * 1e3 / n
warm = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev)
for _ in range(200): task.cam.render_envs(out=warm)
print("first buffer %x: %.1f us" % (warm.data_ptr(), t(warm)))
pool = torch.empty(N + (64 << 20), dtype=torch.uint8, device=dev)
print("pool base %x" % pool.data_ptr())
for rep in range(1):
    for off in (0, 256, 1024, 4096, 8192, 32768, 65536, 1 << 20, (2 << 20) + 4096, 16 << 20, (32 << 20) + 12288):
        out = pool[off:off + N].view(B, H, W, 3)
        print("offset %9d: %.1f us" % (off, t(out)))
bufs = [torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4)]
for b in bufs: print("separate buffer %x: %.1f us" % (b.data_ptr(), t(b)))
print("first buffer again: %.1f us" % t(warm))
