"""Divergence guard (include/mirigid.h: mir_get_bad): an env whose state goes non-finite is flagged while diagnostics
are on, never terminates, and does not touch any other env -- in both step kernels, through the fused launch, the rollout launch and
the begin / end pair with early terminated bytes."""
import numpy as np
import pytest
import torch

from gym_genesis.backend import models

pytestmark = pytest.mark.gpu

HOME = np.array(models.FRANKA_HOME, dtype=np.float32)


def _reset(sc, B):
    rng = np.random.RandomState(0)
    pos = np.stack([rng.uniform(0.45, 0.80, B), rng.uniform(-0.25, 0.25, B), np.full(B, 0.02)], 1).astype(np.float32)
    sc.reset(pos, np.tile(np.array([0, 0, 0, 1], np.float32), (B, 1)), np.tile(HOME, (B, 1)))


@pytest.mark.parametrize("path", ["fused", "begin_end", "rollout"])
def test_non_finite_env_is_flagged_and_leaves_the_other_4095_bit_identical(franka_spec, path):
    """A NaN joint velocity written into ONE env (a diverged env, or a caller's bad state) after five steps; the clean twin gets the
    same state write without it, so that both runs launch the same kernels."""
    from gym_genesis.backend.lib import MirScene

    B, T, victim, t_bad = 4096, 24, 1234, 5
    acts = np.random.default_rng(5).uniform(-1, 1, (T, B, 9)).astype(np.float32)
    outs = []
    for poison in (False, True):
        sc = MirScene(franka_spec, B)
        sc.set_diag(True)
        _reset(sc, B)
        sc.get_bad(reset=True)
        a = torch.as_tensor(acts, device=sc.device)
        b = (sc.empty(9), sc.empty(11), sc.empty(), sc.empty(dtype=torch.uint8))
        host, flagged_at, per = [], [], []

        def inject():
            q, v, tg, ws = sc.get_state()
            if poison:
                v = v.clone(); v[victim, 3] = float("nan")
            sc.set_state(qpos=q, qvel=v, target=tg, warmstart=ws)

        if path == "rollout":
            rows = torch.zeros((T, B, 24), dtype=torch.float32, device=sc.device)
            sc.rollout(a[:t_bad].contiguous(), rows[:t_bad])
            assert not sc.get_bad()[0].any()
            inject()
            sc.rollout(a[t_bad:].contiguous(), rows[t_bad:])
            bad, n = sc.get_bad()
            flagged_at = [int(x) for x in torch.nonzero(bad).flatten().cpu()]
            res = rows[:, :, :22].cpu().numpy()
        else:
            for t in range(T):
                if t == t_bad:
                    inject()
                if path == "fused":
                    sc.step_fused(a[t], *b)
                else:
                    sc.step_begin(a[t], *b); host.append(sc.step_end())
                per.append(torch.cat([b[0], b[1], b[2][:, None], b[3][:, None].float()], 1).cpu().numpy())
                bad, n = sc.get_bad()
                if bad.any():
                    flagged_at.append((t, [int(x) for x in torch.nonzero(bad).flatten().cpu()]))
            res = np.stack(per)
        outs.append((res, res[:, :, 21], flagged_at, n, [x.cpu().numpy() for x in sc.get_state()], host))
    (r0, t0, f0, n0, s0, h0), (r1, t1, f1, n1, s1, h1) = outs
    assert f0 == [] and n0 == 0
    others = np.arange(B) != victim
    assert np.array_equal(r0[:, others], r1[:, others]), "a non-finite env changed another env's outputs"
    for x, y in zip(s0, s1):
        assert np.array_equal(x[others], y[others])
    assert np.array_equal(r0[:t_bad, victim], r1[:t_bad, victim])
    assert not np.isfinite(s1[0][victim]).all()
    # flagged from the step that took the NaN on, the victim only; never terminated afterwards
    if path == "rollout":
        assert f1 == [victim] and n1 == T - t_bad
    else:
        assert [t for t, _ in f1] == list(range(t_bad, T)) and all(e == [victim] for _, e in f1) and n1 == T - t_bad
    assert not t1[t_bad:, victim].any()
    if path == "begin_end":
        assert np.array_equal(np.stack(h1)[:, others], np.stack(h0)[:, others]) and not np.stack(h1)[t_bad:, victim].any()


def test_nan_target_saturates_the_actuator_and_nothing_diverges(franka_spec):
    """A NaN in an ACTION never reaches the state: the position actuator's force is clamped to the joint's force range, and the clamp
    (fmin / fmax) returns the finite bound for a NaN force.  The env is not flagged."""
    from gym_genesis.backend.lib import MirScene

    B = 256
    sc = MirScene(franka_spec, B)
    sc.set_diag(True)
    _reset(sc, B)
    sc.get_bad(reset=True)
    a = torch.as_tensor(np.random.default_rng(0).uniform(-1, 1, (B, 9)).astype(np.float32), device=sc.device)
    a[17, 2] = float("nan")
    b = (sc.empty(9), sc.empty(11), sc.empty(), sc.empty(dtype=torch.uint8))
    for t in range(10):
        sc.step_fused(a, *b)
    bad, n = sc.get_bad()
    q, v, tg, ws = sc.get_state()
    assert not bad.any() and n == 0 and torch.isfinite(q).all() and torch.isfinite(v).all()


def test_inf_height_does_not_terminate_and_guard_is_silent_without_diagnostics(franka_spec):
    from gym_genesis.backend.lib import MirError, MirScene

    B = 64
    sc = MirScene(franka_spec, B)
    _reset(sc, B)
    q, v, tg, ws = sc.get_state()
    q2 = q.clone(); q2[3, 11] = float("inf"); q2[4, 11] = 0.5   # env 3: cube at +Inf; env 4: cube 0.5 m up (terminates)
    sc.set_state(qpos=q2, qvel=v, target=tg, warmstart=ws)
    agent, env, rew, term = sc.get_obs()
    assert term[4].item() == 1 and rew[4].item() == 1.0
    assert term[3].item() == 0 and rew[3].item() == 0.0
    sc.set_diag(False)
    with pytest.raises(MirError):
        sc.get_bad()


def test_wave_kernel_flags_a_non_finite_env():
    from gym_genesis.backend.lib import MirScene

    B = 128
    spec = models.franka_cube_stack_scene().build()
    outs = []
    for poison in (False, True):
        sc = MirScene(spec, B)
        assert sc.kernel == 64
        sc.set_diag(True)
        sc.get_bad(reset=True)
        b = (sc.empty(sc.agent_dim), sc.empty(sc.env_dim), sc.empty(), sc.empty(dtype=torch.uint8))
        a = sc.get_state()[2].clone()
        for t in range(6):
            if t == 2:   # (the clean run gets the same state write, without the NaN)
                q, v, tg, ws = sc.get_state()
                if poison:
                    v = v.clone(); v[77, 1] = float("nan")
                sc.set_state(qpos=q, qvel=v, target=tg, warmstart=ws)
            sc.step_fused(a, *b)
        bad, n = sc.get_bad()
        outs.append(([x.cpu().numpy() for x in sc.get_state()], bad.cpu().numpy(), n, b[3].cpu().numpy()))
    (s0, b0, n0, t0), (s1, b1, n1, t1) = outs
    others = np.arange(B) != 77
    assert not b0.any() and n0 == 0
    assert b1[77] == 1 and b1.sum() == 1 and n1 == 4 and t1[77] == 0
    for x, y in zip(s0, s1):
        assert np.array_equal(x[others], y[others])


def test_guard_is_silent_on_a_scene_whose_qpos_row_is_shorter_than_16_words():
    """One free box on the plane: nq = 7, the 16-lane kernel's qpos row is 8 words and lanes 8 .. 15 of S.qpos are never written.
    With every CU's LDS poisoned with NaN patterns (the autouse fixture) the guard must stay silent (it read all 16)."""
    from gym_genesis.backend import spec as S
    from gym_genesis.backend.lib import MirScene

    sb = S.SceneBuilder()
    sb.add_geom(0, S.GEOM_PLANE)
    sb.add_body("box", 0, pos=(0.0, 0.0, 0.3), jtype=S.JNT_FREE, mass=0.2, inertia=S.box_inertia(0.2, (0.03, 0.03, 0.03)))
    sb.add_geom("box", S.GEOM_BOX, size=(0.03, 0.03, 0.03))
    sb.task = dict(eef_body=1, obj_body=1, grip_dof=(), reward_z=0.1)
    B = 256
    sc = MirScene(sb.build(), B)
    assert sc.kernel == 16 and sc.nq == 7
    sc.set_diag(True)
    sc.get_bad(reset=True)
    for _ in range(60):   # (falls, lands, rests)
        sc.step(1)
        bad, n = sc.get_bad()
        assert not bad.any() and n == 0
    q = sc.get_state()[0]
    assert torch.isfinite(q).all() and (q[:, 2] - 0.03).abs().max() < 5e-3
