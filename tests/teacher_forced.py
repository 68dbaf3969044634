"""Shared by the contact-rich STATE-parity tests (test infrastructure, like oracle/): a recorded episode -- the state behind its reset and
its actions -- replayed by the float64 oracle, and on EVERY step the device scene and the float32 CPU port of the oracle are restarted
from the oracle's state (rounded to float32) and take the same action: one-step errors against float64, quantile by quantile, with the
float32 port as the yardstick for what float32 arithmetic gives on that workload (the bar of tests/test_gpu_parity.py: device <= 1.5 x
port + 2e-6 per quantile).  Rewards / masks are compared bit for bit away from their thresholds; env-steps whose contact count differs
between float32 and float64 from the SAME state (a point exactly at make / break) are counted and not compared."""
import os

import numpy as np
import torch

import orc

QS = (0.5, 0.9, 0.99, 0.999, 0.9999)


def fmt(x):
    return " ".join(f"{np.quantile(x, q):.1e}" for q in QS) + f" max {x.max():.1e}"


def replay(sc, spec, state0, actions, f32, clear_of_threshold, exact=False):
    """sc: the device scene (MirScene of `spec`); state0 = (qpos, qvel, target, warmstart) NumPy; actions: list of (B, nu) NumPy;
    f32: True / "big" (which float32 port); clear_of_threshold(env_state (B, .), ref_oracle) -> bool (B,) envs whose reward may be
    compared.  -> dict(e_dev, e_port, flips, flips_dev, flips_port, rew_skipped, rew_checked, deferred)"""
    n = sc.num_envs
    o, ref, port = orc.Oracle(spec, n), orc.Oracle(spec, n), orc.Oracle(spec, n, f32=f32)
    sc.set_diag(True)
    q0, v0, t0, w0 = state0
    o.write_all(orc.F_QPOS, q0); o.write_all(orc.F_QVEL, v0); o.write_all(orc.F_QACC_WS, w0)
    tg = np.zeros((n, o.nv)); tg[:, o.u_dofs] = t0
    o.write_all(orc.F_TARGET, tg)
    bufs = (sc.empty(sc.agent_dim), sc.empty(sc.env_dim), sc.empty(), sc.empty(dtype=torch.uint8))
    nt = max(1, min(64, len(os.sched_getaffinity(0))))
    e_dev, e_port, was_def = [], [], []
    flips = flips_dev = flips_port = rew_skipped = rew_checked = 0
    for a in actions:
        q, v = o.state()
        ws, tgt = o.read_all(orc.F_QACC_WS, o.nv), o.read_all(orc.F_TARGET, o.nv)
        q32, v32, w32 = q.astype(np.float32), v.astype(np.float32), ws.astype(np.float32)
        sc.set_state(qpos=q32, qvel=v32, warmstart=w32)
        for x in (ref, port):
            x.write_all(orc.F_QPOS, q32); x.write_all(orc.F_QVEL, v32); x.write_all(orc.F_QACC_WS, w32); x.write_all(orc.F_TARGET, tgt)
        if exact:
            sc.step_begin(torch.as_tensor(a, device=sc.device), *bufs); host = sc.step_end()
            assert np.array_equal(host, bufs[3].cpu().numpy().astype(bool))
        else:
            sc.step_fused(torch.as_tensor(a, device=sc.device), *bufs)
        for x in (o, ref, port):
            x.step_batch(a, nt)
        qo = ref.state()[0]
        dg = sc.get_diag(points=True)
        nd, no, npt = dg[0].cpu().numpy(), ref.counts_all()[0], port.counts_all()[0]
        same = (nd == no) & (no == npt)
        flips += int((~same).sum()); flips_dev += int((nd != no).sum()); flips_port += int((npt != no).sum())
        qh = sc.get_state()[0].cpu().numpy()
        e_dev.append(np.abs(qh - qo).max(1)[same]); e_port.append(np.abs(port.state()[0] - qo).max(1)[same])
        was_def.append(dg[3].cpu().numpy()[same])
        ag, es, ro, to = ref.get_obs_all()
        clear = clear_of_threshold(es, ref)
        rew_skipped += int((~clear).sum()); rew_checked += int(clear.sum())
        assert np.array_equal(bufs[2].cpu().numpy()[clear], ro.astype(np.float32)[clear]), "rewards differ from the oracle's away from the threshold"
        assert np.array_equal(bufs[3].cpu().numpy().astype(bool)[clear], to.astype(bool)[clear]), "terminated differs from the oracle's away from the threshold"
    return dict(e_dev=np.concatenate(e_dev), e_port=np.concatenate(e_port), points=np.concatenate(was_def), flips=flips, flips_dev=flips_dev, flips_port=flips_port,
                rew_skipped=rew_skipped, rew_checked=rew_checked, steps=len(actions), n=n)


def assert_within_float32(r, sel=None):
    e_dev, e_port = (r["e_dev"], r["e_port"]) if sel is None else (r["e_dev"][sel], r["e_port"][sel])
    for qn in QS:
        assert np.quantile(e_dev, qn) <= 1.5 * np.quantile(e_port, qn) + 2e-6, (qn, np.quantile(e_dev, qn), np.quantile(e_port, qn))
