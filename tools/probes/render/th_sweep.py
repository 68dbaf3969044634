"""Strip-height sweep of the binned pixel kernel at BASELINE cfg 5 (1024 x 480 x 640), and the generic kernel beside it."""
import os, sys, torch, numpy as np
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
from gym_genesis.backend import models
from gym_genesis.backend.lib import MirScene
from gym_genesis.backend.spec import make_camera
B = 1024
b = models.franka_cube_pick_scene()
sc = MirScene(b.build(), B)
rng = np.random.RandomState(0)
pos = np.stack([rng.uniform(0.45, 0.80, B), rng.uniform(-0.25, 0.25, B), np.full(B, 0.02)], 1).astype(np.float32)
sc.reset(pos, np.tile(np.array([0, 0, 0, 1], np.float32), (B, 1)), np.tile(np.array(models.FRANKA_HOME, np.float32), (B, 1)))
sc.step(5)
cam = make_camera(640, 480, (3.5, 0, 2.5), (0, 0, 0.5), 30); vis = b.visual()
out = torch.empty((B, 480, 640, 3), dtype=torch.uint8, device=sc.device)
def t(n=20):
    for _ in range(3): sc.render(cam, vis, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): sc.render(cam, vis, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
sc.debug_render_path(generic=True)
print(f"generic: {t():.1f} us")
for rows in (32, 64, 96, 160, 256, 480):
    sc.debug_render_path(generic=False, strip_rows=rows)
    us = t()
    print(f"binned strip {rows}: {us:.1f} us  {B*480*640*3/us/1e6:.2f} TB/s")
cam = make_camera(640, 480, (3.5, 0, 2.5), (0, 4.0, 0.5), 30)
sc.debug_render_path(generic=False, strip_rows=0)
print(f"binned, robot out of view (floor only): {t():.1f} us")
sc.debug_render_path(generic=True)
print(f"generic, robot out of view (floor only): {t():.1f} us")
# the bench's state: arms apart after 20 random-action steps
import torch as _t
g = _t.Generator(device=sc.device).manual_seed(99)
for _ in range(20):
    sc.set_pd_targets(_t.empty((B, 9), dtype=_t.float32, device=sc.device).uniform_(-1.0, 1.0, generator=g)); sc.step(1)
cam = make_camera(640, 480, (3.5, 0, 2.5), (0, 0, 0.5), 30)
sc.debug_render_path(generic=False, strip_rows=0)
print(f"binned, arms apart (the bench's state): {t():.1f} us")
