import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import orc
from gym_genesis.backend import models
from gym_genesis.backend.lib import MirScene
G_ = json.load(open(os.path.join(ROOT, "tests/golden/grasp_targets.json")))
T = np.array(G_["targets"], np.float32); pos = np.array([[x, y, 0.02] for x, y in G_["cube_xy"]], np.float32)
acts = np.repeat(T.transpose(1, 0, 2), G_["steps_per_stage"], axis=0); B = pos.shape[0]
spec = models.franka_cube_pick_scene().build()
sc, o = MirScene(spec, B), orc.Oracle(spec, B)
quat = np.tile(np.array([0,0,0,1],np.float32),(B,1)); arm = np.tile(np.array(models.FRANKA_HOME,np.float32),(B,1))
sc.reset(pos,quat,arm); o.reset(pos,quat,arm)
for t in range(acts.shape[0]):
    sc.set_pd_targets(acts[t]); sc.step(1); o.step_batch(acts[t])
    q, v, _, _ = (x.cpu().numpy() for x in sc.get_state()); qo, vo = o.state()
    nc, ne, ni = (x.cpu().numpy() for x in sc.get_diag()); oc = [o.counts(e) for e in range(B)]
    eq = np.abs(q-qo).max(1); ev = np.abs(v-vo).max(1)
    mism = [int(nc[e] != oc[e][0]) for e in range(B)]
    if t % 10 == 9 or any(mism) or eq.max() > 2e-5:
        print(f"t={t} eq {np.array2string(eq,precision=1)} ev {np.array2string(ev,precision=1)} ncon hip {nc.tolist()} orc {[c[0] for c in oc]} niter hip {ni.tolist()} orc {[c[2] for c in oc]}")
