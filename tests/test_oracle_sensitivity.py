"""The BASELINE random-action workload is chaotic: documents (and pins) why long free-running
float32-vs-float64 comparisons cannot hold 1e-4 on that workload, for any implementation.
CPU only (oracle vs perturbed oracle)."""
import numpy as np

import orc
from gym_genesis.backend import models


def test_random_action_rollout_amplifies_1e7_perturbation(franka_spec):
    B, T = 16, 600
    oa, ob = orc.Oracle(franka_spec, B), orc.Oracle(franka_spec, B)
    rng = np.random.RandomState(1)
    pos = np.stack([rng.uniform(.45, .8, B), rng.uniform(-.25, .25, B), np.full(B, .02)], 1).astype(np.float32)
    quat = np.tile(np.array([0, 0, 0, 1], np.float32), (B, 1))
    arm = np.tile(np.array(models.FRANKA_HOME, np.float32), (B, 1))
    oa.reset(pos, quat, arm)
    ob.reset(pos, quat, arm)
    for e in range(B):  # perturb by about one float32 ulp
        q = ob.read(orc.F_QPOS, e)
        q[:7] += 1e-7 * np.cos(np.arange(7) + e)
        ob.write(orc.F_QPOS, q, e)
    acts = np.random.default_rng(1234).uniform(-1, 1, (T, B, 9)).astype(np.float32)
    err = {}
    for t in range(T):
        oa.step_batch(acts[t])
        ob.step_batch(acts[t])
        if t + 1 in (100, 600):
            err[t + 1] = np.abs(oa.state()[0] - ob.state()[0]).max()
    assert err[100] < 1e-4          # short horizons are comparable ...
    assert err[600] > 1e-4          # ... long ones are not: float64 vs float64 already exceeds the bar
    assert err[600] > 50 * err[100]  # exponential growth
