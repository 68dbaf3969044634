import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gym-genesis_amd")]
import numpy as np, torch
from gym_genesis.backend import models
from gym_genesis.backend.lib import MirScene
for which, cap in (("stack", 8), ("stack", 20), ("pick", 4), ("pick", 7)):
    sb = models.franka_cube_stack_scene() if which == "stack" else models.franka_cube_pick_scene()
    sb.opt["max_contacts"] = cap
    sc = MirScene(sb.build(), 256)
    B = 256; nfree = 5 if which == "stack" else 1
    rng = np.random.RandomState(1)
    if which == "stack":
        from gym_genesis.env import GenesisEnv
        pos = np.zeros((B, nfree, 3), np.float32); pos[..., 0] = rng.uniform(-0.3, 0.3, (B, nfree)); pos[..., 1] = rng.uniform(-0.25, 0.25, (B, nfree)); pos[..., 2] = models.STACK_CUBE_Z
        quat = np.tile(np.array([1, 0, 0, 0], np.float32), (B, nfree, 1))
    else:
        pos = np.stack([rng.uniform(.45,.8,B), rng.uniform(-.25,.25,B), np.full(B,.02)],1).astype(np.float32); quat = np.tile(np.array([1,0,0,0],np.float32),(B,1))
    home = np.tile(np.array(models.FRANKA_HOME, np.float32), (B, 1))
    sc.reset(pos, quat, home)
    g = torch.Generator(device=sc.device).manual_seed(0)
    bufs = (sc.empty(sc.agent_dim), sc.empty(sc.env_dim), sc.empty(), sc.empty(dtype=torch.uint8))
    for t in range(300):
        a = torch.from_numpy(home).to(sc.device) + torch.empty((B, 9), device=sc.device).uniform_(-1.5, 1.5, generator=g)
        sc.step_fused(a, *bufs)
    torch.cuda.synchronize()
    q, v, _, _ = sc.get_state(); nc = sc.get_diag()[0]
    print(which, "cap", cap, "finite", bool(torch.isfinite(q).all() and torch.isfinite(v).all()), "max ncon", int(nc.max()), "hits", int((nc >= cap).sum()))
