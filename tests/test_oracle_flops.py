"""The instrumented build of the oracle (oracle/liborc_flops.so: orc_rigid.c compiled as C++ with an arithmetic type that counts
its own operations, oracle/orc_flops.h) -- the source of bench.py's F_step for the fp32-vector roofline.  It must be
the same computation as the float32 port it is compiled from, count deterministically, and count what a hand count gives on a case
small enough to count by hand."""
import numpy as np

import orc
from gym_genesis.backend import models

HOME = np.array(models.FRANKA_HOME, np.float32)


def _setup(o, B):
    rng = np.random.RandomState(0)
    pos = np.stack([rng.uniform(0.45, 0.80, B), rng.uniform(-0.25, 0.25, B), np.full(B, 0.02)], 1).astype(np.float32)
    o.reset(pos, np.tile(np.array([0, 0, 0, 1.0], np.float32), (B, 1)), np.tile(HOME, (B, 1)))


def test_counted_build_is_the_float32_port_and_counts_deterministically(franka_spec):
    B, T = 8, 30
    acts = np.random.default_rng(3).uniform(-1, 1, (T, B, 9)).astype(np.float32)
    port, cnt = orc.Oracle(franka_spec, B, f32=True), orc.Oracle(franka_spec, B, f32=True, flops=True)
    totals = []
    for rep in range(2):
        _setup(port, B)
        _setup(cnt, B)
        cnt.flops_reset()
        for t in range(T):
            port.step_batch(acts[t], 1)
            cnt.step_batch(acts[t], 1)
        totals.append(cnt.flops())
        qp, vp = port.state()
        qc, vc = cnt.state()
        # same source, same float32 arithmetic; the timed port is built with -march=native (fused multiply-adds), the counted one not
        assert np.abs(qp - qc).max() < 1e-4 and np.abs(vp - vc).max() < 1e-2
    assert totals[0] == totals[1]
    f = totals[0]
    per = sum(v for k, v in f.items() if k != "cmp") / (B * T)
    assert 15e3 < per < 120e3, per          # SURVEY.md 8a-11 estimated 30-100 kflop per env-step
    assert f["mul"] > f["div"] > f["sqrt"] > 0 and f["trans"] > 0
    assert port.flops() == dict.fromkeys(orc.Oracle.FLOP_KINDS, 0)   # the plain builds count nothing


def test_hand_count_of_a_quaternion_product(franka_spec):
    """orc_fk on a scene composes one quaternion product and one rotation per body and geom; its count must scale exactly with the
    number of calls (no hidden state), and a single call costs at least the 16 multiplications + 12 additions of each of the
    nbody - 1 quaternion products."""
    o = orc.Oracle(franka_spec, 1, f32=True, flops=True)
    _setup(o, 1)
    o.flops_reset()
    o.fk(0)
    one = o.flops()
    o.flops_reset()
    for _ in range(5):
        o.fk(0)
    five = o.flops()
    assert all(five[k] == 5 * one[k] for k in one)
    nb = franka_spec.nbody - 1
    assert one["mul"] >= 16 * nb and one["add"] >= 12 * nb
