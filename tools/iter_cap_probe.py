"""Developer probe: launch time vs a cap on Newton iterations (tests the straggler-tail hypothesis) + niter histogram."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]
from gym_genesis.backend import models
from gym_genesis.backend.lib import MirScene
B = 4096
for cap in (50, 3, 2, 1):
    sb = models.franka_cube_pick_scene(); sb.opt["iterations"] = cap
    sc = MirScene(sb.build(), B)
    rng = np.random.RandomState(0)
    pos = np.stack([rng.uniform(.45,.8,B), rng.uniform(-.25,.25,B), np.full(B,.02)],1).astype(np.float32)
    sc.reset(pos, np.tile(np.array([0,0,0,1],np.float32),(B,1)), np.tile(np.array(models.FRANKA_HOME,np.float32),(B,1)))
    g = torch.Generator(device=sc.device).manual_seed(1)
    acts = torch.empty((64,B,9),device=sc.device).uniform_(-1,1,generator=g)
    bufs = (sc.empty(9), sc.empty(11), sc.empty(), sc.empty(dtype=torch.uint8))
    hist = np.zeros(8)
    for t in range(60):
        sc.step_fused(acts[t%64], *bufs)
        if t >= 30: hist += np.bincount(np.minimum(sc.get_diag()[2].cpu().numpy(), 7), minlength=8)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(400): sc.step_fused(acts[t%64], *bufs)
    torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/400
    print(f"iterations cap {cap:2d}: {dt*1e6:6.1f} us/step   niter histogram (per env-step): {np.round(hist/hist.sum(),4)}")
