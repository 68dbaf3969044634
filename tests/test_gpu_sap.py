"""GPU tests of the sweep-and-prune broadphase (16-lane kernel, FEAT bit 1): world AABBs per geom, sort by lo.x, sweep, static
pair filter as per-geom bit masks, survivors in the order of the static list.

  * forced on the benchmark scene (MIR_BROADPHASE=sap) it must reproduce the static-list run BIT FOR BIT: both feed the same
    bounding tests and the same narrowphase, in the same order;
  * a scene whose static list would not fit (two compound bodies of ten spheres each: 120 pairs > K16_MAX_PAIR = 64) runs on
    the sweep automatically and is held against the float64 oracle, which walks all pairs.
"""
import json
import os

import numpy as np
import pytest
import torch

import orc
from gym_genesis.backend import models

pytestmark = pytest.mark.gpu
HOME = np.array(models.FRANKA_HOME, dtype=np.float32)


def _mir(spec, B, broadphase=None):
    from gym_genesis.backend.lib import MirScene

    old = os.environ.pop("MIR_BROADPHASE", None)
    if broadphase:
        os.environ["MIR_BROADPHASE"] = broadphase
    try:
        return MirScene(spec, B)
    finally:
        os.environ.pop("MIR_BROADPHASE", None)
        if old is not None:
            os.environ["MIR_BROADPHASE"] = old


def _reset(scs, B, seed):
    rng = np.random.RandomState(seed)
    pos = np.stack([rng.uniform(0.45, 0.80, B), rng.uniform(-0.25, 0.25, B), np.full(B, 0.02)], 1).astype(np.float32)
    quat = np.tile(np.array([0, 0, 0, 1], np.float32), (B, 1))
    for s in scs:
        s.reset(pos, quat, np.tile(HOME, (B, 1)))


def test_sap_reproduces_the_static_list_bit_for_bit(franka_spec):
    B = 256
    a, b = _mir(franka_spec, B, "static"), _mir(franka_spec, B, "sap")
    _reset((a, b), B, 0)
    acts = torch.as_tensor(np.random.default_rng(4).uniform(-1, 1, (150, B, 9)).astype(np.float32), device=a.device)
    ba = (a.empty(9), a.empty(11), a.empty(), a.empty(dtype=torch.uint8))
    bb = (b.empty(9), b.empty(11), b.empty(), b.empty(dtype=torch.uint8))
    for t in range(150):
        a.step_fused(acts[t], *ba)
        b.step_fused(acts[t], *bb)
    for x, y in zip(a.get_state(), b.get_state()):
        assert torch.equal(x, y)
    for x, y in zip(ba, bb):
        assert torch.equal(x, y)
    assert torch.equal(a.get_diag()[0], b.get_diag()[0])
    # the scripted grasp: finger pads, cube, plane -- contacts every step
    G_ = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "grasp_targets.json")))
    T = torch.as_tensor(np.array(G_["targets"], np.float32), device=a.device)
    n = T.shape[0]
    a, b = _mir(franka_spec, n, "static"), _mir(franka_spec, n, "sap")
    cube = np.array([[x, y, 0.02] for x, y in G_["cube_xy"]], np.float32)
    for s in (a, b):
        s.reset(cube, np.tile(np.array([0, 0, 0, 1], np.float32), (n, 1)), np.tile(HOME, (n, 1)))
    ba = (a.empty(9), a.empty(11), a.empty(), a.empty(dtype=torch.uint8))
    bb = (b.empty(9), b.empty(11), b.empty(), b.empty(dtype=torch.uint8))
    maxcon = 0
    for s in range(T.shape[1]):
        for _ in range(G_["steps_per_stage"]):
            a.step_fused(T[:, s].contiguous(), *ba)
            b.step_fused(T[:, s].contiguous(), *bb)
            maxcon = max(maxcon, int(a.get_diag()[0].max()))
    for x, y in zip(a.get_state(), b.get_state()):
        assert torch.equal(x, y)
    assert bool(ba[3].all()) and maxcon >= 8


def _bead_scene():
    """plane + a static chain of ten small spheres 20 cm above it + one free body made of ten spheres (a compound shape):
    10 x 10 bead pairs + 10 plane pairs = 110 candidate pairs, more than the static list holds (K16_MAX_PAIR = 64)."""
    from gym_genesis.backend import spec as S

    sb = S.SceneBuilder()
    sb.add_geom(0, S.GEOM_PLANE)
    for k in range(10):
        sb.add_geom(0, S.GEOM_SPHERE, size=(0.02, 0.0, 0.0), pos=(0.03 * (k - 4.5), 0.0, 0.2))
    sb.add_body("b", 0, pos=(0.0, 0.0, 0.5), jtype=S.JNT_FREE, mass=0.2, inertia=S.box_inertia(0.2, (0.14, 0.02, 0.02)))
    for k in range(10):
        sb.add_geom("b", S.GEOM_SPHERE, size=(0.02, 0.0, 0.0), pos=(0.03 * (k - 4.5), 0.0, 0.0))
    sb.task = dict(eef_body=1, obj_body=1, grip_dof=(), reward_z=0.1)
    return sb.build()


def test_long_pair_list_runs_on_the_sweep_and_matches_the_oracle():
    """The free bead chain dropped crosswise on the static one: it lands on one or two beads, rocks, slides off and ends on the
    plane.  mir_create switches to the sweep (the static list would not fit); the oracle walks all 110 pairs.  Contact counts
    agree, positions within 1e-4 for 90 % of the envs after 60 steps (before the chaotic slide-off)."""
    spec = _bead_scene()
    B = 128
    sc, o = _mir(spec, B), orc.Oracle(spec, B)
    assert sc.kernel == 16 and sc.npair == 110
    rng = np.random.default_rng(3)
    pos = np.zeros((B, 1, 3), np.float32)
    pos[:, 0] = rng.uniform(-0.03, 0.03, (B, 3)) + [0.0, 0.0, 0.27]
    ang = rng.uniform(0.9, 2.2, B)                       # the free chain lies across the static one
    quat = np.zeros((B, 1, 4), np.float32)
    quat[:, 0, 0], quat[:, 0, 3] = np.cos(ang / 2), np.sin(ang / 2)
    arm = np.zeros((B, 0), np.float32)
    sc.reset(pos, quat, arm)
    o.reset(pos, quat, arm)
    bufs = (sc.empty(sc.agent_dim), sc.empty(sc.env_dim), sc.empty(), sc.empty(dtype=torch.uint8))
    agree, seen, err60 = [], 0, None
    for t in range(200):
        sc.step_fused(None, *bufs)
        o.step_batch(None)
        if t % 10 == 9:
            nc = sc.get_diag()[0].cpu().numpy()
            nco = np.array([o.counts(e)[0] for e in range(B)])
            agree.append(float((nc == nco).mean()))
            seen = max(seen, int(nco.max()))
        if t == 59:
            err60 = np.abs(sc.get_state()[0].cpu().numpy() - o.state()[0])[:, :3].max(1)
    zf = sc.get_state()[0].cpu().numpy()[:, 2]
    print(f"bead chains: 110 candidate pairs (sweep-and-prune), up to {seen} contacts per env, contact counts agree in {min(agree[:6]) * 100:.0f} % of "
          f"the envs over the first 60 steps, position err at step 60 median {np.median(err60):.2e}, 90 % {np.quantile(err60, 0.9):.2e}; "
          f"{int((zf < 0.05).sum())} of {B} chains ended on the plane")
    assert seen >= 2 and min(agree[:6]) > 0.95
    assert np.quantile(err60, 0.9) < 1e-4
    assert (zf < 0.05).sum() > B // 2                    # most chains have slid off and lie on the plane (ten plane contacts)
