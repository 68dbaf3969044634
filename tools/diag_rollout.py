"""Diagnostic: error trajectory of the HIP path and of the float32 oracle port against the float64 oracle."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd"), os.path.join(ROOT, "tests")]
import orc
from gym_genesis.backend import models
from gym_genesis.backend.lib import MirScene

B, T = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
spec = models.franka_cube_pick_scene().build()
sc = MirScene(spec, B); o64 = orc.Oracle(spec, B); o32 = orc.Oracle(spec, B, f32=True)
rng = np.random.RandomState(1)
pos = np.stack([rng.uniform(.45,.8,B), rng.uniform(-.25,.25,B), np.full(B,.02)],1).astype(np.float32)
quat = np.tile(np.array([0,0,0,1],np.float32),(B,1)); arm = np.tile(np.array(models.FRANKA_HOME,np.float32),(B,1))
for s in (sc, o64, o32): s.reset(pos, quat, arm)
acts = np.random.default_rng(1234).uniform(-1,1,(T,B,9)).astype(np.float32)
for t in range(T):
    sc.set_pd_targets(acts[t]); sc.step(1); o64.step_batch(acts[t]); o32.step_batch(acts[t])
    if t % 50 == 49 or t < 3:
        q, v, _, _ = (x.cpu().numpy() for x in sc.get_state())
        q64, v64 = o64.state(); q32, v32 = o32.state()
        eq = np.abs(q-q64); e32 = np.abs(q32-q64)
        nc, ne, ni = (x.cpu().numpy() for x in sc.get_diag())
        print(f"t={t+1:4d} hip: q {eq.max():.2e} (dof {eq.max(0).argmax()}, env {eq.max(1).argmax()}) v {np.abs(v-v64).max():.2e} | orc32: q {e32.max():.2e} v {np.abs(v32-v64).max():.2e} | ncon {nc.min()}-{nc.max()} nefc {ne.min()}-{ne.max()} niter {ni.min()}-{ni.max()} mean {ni.mean():.2f}")
print("per-dof hip max err", np.round(eq.max(0),7))
