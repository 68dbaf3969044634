"""The stopping rules the float64 oracle shares with the float32 kernels BY CONSTRUCTION: the rounding
floor of the gradient and the stagnation rule built on it (oracle/orc_rigid.c: gfloor), the rounding floor of the line search's phi'
"from the fifth evaluation on", and the relative tolerance of the search (LS_RELTOL).  A modelling mistake in one of them would be
invisible to every GPU parity test, because both sides would make it.  In float64 none of them should ever bind before the
solver's tolerance does.  Shown here: the same contact-rich states -- the scripted grasp: fingers on the floor, pads on the cube,
up to 30 contact points -- solved by the oracle as it is and by liborc64_norules.so (-DORC_NO_RULES: the rules compiled out), at
the default tolerance and at 1e-14: the same acceleration to 1e-9 (relative to the largest component).

What this does NOT cover: thin_manifolds (a modelling choice at capacity 16, nothing to be insensitive to -- the capacity study of
tests/test_ref_expert.py and the exact-contacts path of tests/test_gpu_exact_contacts.py are its checks), and the rules' float32
values in the kernels (those are what the float32 yardstick tests are for)."""
import json
import os

import numpy as np

import orc
from gym_genesis.backend import models

HOME = np.array(models.FRANKA_HOME, dtype=np.float32)


def _spec(tol=None):
    sb = models.franka_cube_pick_scene()
    sb.opt["max_contacts"] = 48
    if tol is not None:
        sb.opt["tolerance"] = tol
        sb.opt["iterations"] = 200
        sb.opt["ls_iterations"] = 200
    return sb.build()


def test_float64_result_is_insensitive_to_the_rules_shared_with_the_kernels():
    G_ = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "grasp_targets.json")))
    T = np.array(G_["targets"], np.float32)
    B = 16
    pos = np.tile(np.array([[x, y, 0.02] for x, y in G_["cube_xy"]], np.float32), (B // 4, 1))
    pos[:, :2] += np.random.default_rng(3).uniform(-0.002, 0.002, (B, 2)).astype(np.float32)
    acts = np.tile(np.repeat(T.transpose(1, 0, 2), G_["steps_per_stage"], axis=0), (1, B // 4, 1))
    quat = np.tile(np.array([0, 0, 0, 1], np.float32), (B, 1))
    run = orc.Oracle(_spec(), B)                     # the trajectory that supplies the states
    pairs = [(orc.Oracle(_spec(t), B), orc.Oracle(_spec(t), B, variant="norules")) for t in (None, 1e-14)]
    run.reset(pos, quat, np.tile(HOME, (B, 1)))
    worst, ncon_max, n_states, it_max = [0.0, 0.0], 0, 0, 0
    for t in range(acts.shape[0]):
        if t % 4 == 0:
            q, v = run.state()
            ws, tg = run.read_all(orc.F_QACC_WS, run.nv), acts[t]
            for k, (a, b) in enumerate(pairs):
                for o in (a, b):
                    o.write_all(orc.F_QPOS, q); o.write_all(orc.F_QVEL, v); o.write_all(orc.F_QACC_WS, ws); o.set_targets(tg)
                for e in range(B):
                    a.forward(e); b.forward(e)
                    qa, qb = a.read(orc.F_QACC, e), b.read(orc.F_QACC, e)
                    worst[k] = max(worst[k], float(np.abs(qa - qb).max() / max(1.0, np.abs(qb).max())))
                    ncon_max = max(ncon_max, a.counts(e)[0])
                    it_max = max(it_max, b.counts(e)[2])
                    assert a.counts(e)[0] == b.counts(e)[0]
            n_states += B
        run.step_batch(acts[t], 1)
    print(f"\n[oracle rules] {n_states} states of the scripted grasp (up to {ncon_max} contact points): qacc with the shared stopping rules against "
          f"without them, relative L-inf: {worst[0]:.2e} at the default tolerance, {worst[1]:.2e} at tolerance 1e-14 (up to {it_max} Newton iterations)")
    assert ncon_max >= 20
    assert worst[0] < 1e-9 and worst[1] < 1e-9
