import os, sys, time, torch, numpy as np
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "gym-genesis_amd"))
from gym_genesis.backend import models
from gym_genesis.backend.lib import MirScene
from gym_genesis.backend.spec import make_camera
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = models.franka_cube_pick_scene()
sc = MirScene(b.build(), B)
rng = np.random.RandomState(0)
pos = np.stack([rng.uniform(0.45, 0.80, B), rng.uniform(-0.25, 0.25, B), np.full(B, 0.02)], 1).astype(np.float32)
sc.reset(pos, np.tile(np.array([0, 0, 0, 1], np.float32), (B, 1)), np.tile(np.array(models.FRANKA_HOME, np.float32), (B, 1)))
sc.step(5)
for (W, H) in ((640, 480), (128, 96)):
    cam = make_camera(W, H, (3.5, 0, 2.5), (0, 0, 0.5), 30); vis = b.visual()
    out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=sc.device)
    for _ in range(3): sc.render(cam, vis, out=out)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    ev0.record()
    for _ in range(n): sc.render(cam, vis, out=out)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / n
    gb = B * H * W * 3 / 1e9
    print(f"B={B} {W}x{H}: {ms*1e3:.1f} us/render  {gb/ms*1e3:.0f} GB/s written  {B/ms*1e3:.0f} env-frames/s")
    if len(sys.argv) > 2:  # (`render_time.py 1024 floor`; the profiling runs leave it out)
        # the same view moved 5 m sideways: the floor alone (no box in view), i.e. what the floor pass and the stores cost
        camf = make_camera(W, H, (3.5, 5.0, 2.5), (0, 5.0, 0.5), 30)
        for _ in range(3): sc.render(camf, vis, out=out)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(n): sc.render(camf, vis, out=out)
        ev1.record(); torch.cuda.synchronize()
        print(f"    floor alone (robot out of view): {ev0.elapsed_time(ev1) / n * 1e3:.1f} us/render")
    # calibration: a plain device fill of the same buffer (what the write path sustains; WRITE_SIZE per known byte count)
    flat = out.view(-1).view(torch.int32)
    for _ in range(3): flat.fill_(7)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(n): flat.fill_(7)
    ev1.record(); torch.cuda.synchronize()
    msf = ev0.elapsed_time(ev1) / n
    print(f"    torch fill_ of the same {gb*1e3:.0f} MB: {msf*1e3:.1f} us  {gb/msf*1e3:.0f} GB/s")
