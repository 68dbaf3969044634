"""Developer probe: step time and Newton-iteration histogram in windows of a long random-action rollout."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]
from gym_genesis.backend import models
from gym_genesis.backend.lib import MirScene
B = 4096
sc = MirScene(models.franka_cube_pick_scene().build(), B)
rng = np.random.RandomState(0)
pos = np.stack([rng.uniform(.45,.8,B), rng.uniform(-.25,.25,B), np.full(B,.02)],1).astype(np.float32)
sc.reset(pos, np.tile(np.array([0,0,0,1],np.float32),(B,1)), np.tile(np.array(models.FRANKA_HOME,np.float32),(B,1)))
g = torch.Generator(device=sc.device).manual_seed(1234)
acts = torch.empty((2048,B,9),device=sc.device).uniform_(-1,1,generator=g)
bufs = (sc.empty(9), sc.empty(11), sc.empty(), sc.empty(dtype=torch.uint8))
W = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for w in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mx = torch.zeros((), dtype=torch.int32, device=sc.device); big = torch.zeros((), dtype=torch.int64, device=sc.device); mc = torch.zeros((), dtype=torch.int32, device=sc.device)
    for t in range(W):
        sc.step_fused(acts[(w*W+t) % 2048], *bufs)
        d = sc.get_diag(); mx = torch.maximum(mx, d[2].max()); big += (d[2] >= 6).sum(); mc = torch.maximum(mc, d[0].max())
    torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/W
    nc, ne, ni = (x.cpu().numpy() for x in sc.get_diag())
    hist = np.bincount(np.minimum(ni,7), minlength=8)/B
    q, v, _, _ = sc.get_state(); print(f"steps {w*W:4d}-{w*W+W-1}: |qvel|max {float(v.abs().max()):.1f} mean {float(v.abs().mean()):.2f} {dt*1e6:6.1f} us/step  niter hist {np.round(hist[1:6],4)}  window: niter max {int(mx)} #(niter>=6) {int(big)} ncon max {int(mc)}  nefc mean {ne.mean():.1f}")
