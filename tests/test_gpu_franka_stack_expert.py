"""The reference's other Franka caller, verbatim: the batched stack expert of /root/reference/examples/franka/stack_cube_state.py:18-95,139-147
(examples/franka/stack_cube_state.py restates it constant for constant: Cartesian waypoints from `get_link("hand").get_pos(envs_idx=)`, IK
chained through `init_qpos` from `get_qpos(envs_idx=)`, 80 // (waypoints - 1) interpolated joint targets per pair, fingers 0.04 / -0.02) on
CubeStack-v0 with the Franka (39 dofs, five cubes, the wave-per-env kernel) -- the only behavioural fixture the reference holds for that
task.  Free-running, the device's verdicts against the oracle's env by env (as tests/test_ref_expert.py does for the
pick expert); and every one of its 391 steps teacher-forced from the float64 oracle, the float32 CPU port as yardstick."""
import importlib.util
import os

import numpy as np
import pytest

import teacher_forced as tf

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _expert():
    spec = importlib.util.spec_from_file_location("stack_cube_state", os.path.join(ROOT, "examples", "franka", "stack_cube_state.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _stack_clear(es, ref):
    dxy = np.hypot(es[:, 0] - es[:, 11], es[:, 1] - es[:, 12]); dz = es[:, 2] - es[:, 13]
    return (np.abs(dxy - 0.05) > 1e-5) & (np.abs(dz - 0.03) > 1e-5)


def test_reference_franka_stack_expert_verdicts_device_equals_oracle(monkeypatch):
    import fake_scene
    from gym_genesis.env import GenesisEnv
    from gym_genesis.tasks import stack_common

    ex = _expert()
    assert ex.STAGES == ("hover", "grasp", "lift", "place", "release")
    n = 128

    def episode():
        env = GenesisEnv(task="cube_stack", robot="franka", num_envs=n)
        obs, _ = env.reset(seed=2)
        spawn = np.asarray(obs["environment_state"][:, :3].cpu())
        states, envs, acts, rews = ex.run_episode(env, obs)
        return spawn, envs, rews

    s_dev, e_dev, r_dev = episode()
    assert r_dev.shape == (391, n) and np.isfinite(e_dev).all()
    monkeypatch.setattr(stack_common, "MirScene", fake_scene.OracleScene)
    s_orc, e_orc, r_orc = episode()
    assert np.allclose(s_dev, s_orc, atol=1e-5)
    ok_dev, ok_orc = (r_dev > 0).any(axis=0), (r_orc > 0).any(axis=0)   # any positive reward (:156)
    end_dev, end_orc = r_dev[-1] > 0, r_orc[-1] > 0
    print(f"\n[reference stack expert (Franka), free-running x {n}, 391 steps] cube_1 on cube_2 at some step: device {ok_dev.mean():.3f}, oracle {ok_orc.mean():.3f}, "
          f"same verdict in {np.mean(ok_dev == ok_orc):.3f} of the envs; at the end: device {end_dev.mean():.3f}, oracle {end_orc.mean():.3f}, same in {np.mean(end_dev == end_orc):.3f}")
    assert ok_orc.mean() >= 0.5 and abs(ok_dev.mean() - ok_orc.mean()) <= 0.03 and np.mean(ok_dev == ok_orc) >= 0.95
    assert abs(end_dev.mean() - end_orc.mean()) <= 0.05 and np.mean(end_dev == end_orc) >= 0.9


def test_reference_franka_stack_expert_teacher_forced_state_parity():
    from gym_genesis.backend.lib import MirScene
    from gym_genesis.env import GenesisEnv

    ex = _expert()
    n = 128
    env = GenesisEnv(task="cube_stack", robot="franka", num_envs=n)
    obs, _ = env.reset(seed=2)
    rec = {}
    states, envs, acts, rews = ex.run_episode(env, obs, record=rec)
    sc = MirScene(rec["spec"], n)
    assert sc.kernel == 64
    r = tf.replay(sc, rec["spec"], rec["state0"], rec["actions"], "big", _stack_clear)
    print(f"\n[Franka stack scene, the reference's expert x {n}, 391 steps, stacked at some step {np.mean((rews > 0).any(0)):.2f}] one-step qpos L-inf, quantiles {tf.QS}: "
          f"device {tf.fmt(r['e_dev'])} | float32 CPU port {tf.fmt(r['e_port'])}; contact-count flips excluded {r['flips']} of {391 * n} (device vs oracle "
          f"{r['flips_dev']}, port vs oracle {r['flips_port']}); rewards compared {r['rew_checked']}, at a threshold {r['rew_skipped']}; points mean {r['points'].mean():.1f} max {r['points'].max()}")
    assert np.mean((rews > 0).any(0)) > 0.5
    assert r["flips"] < 391 * n // 8 and r["flips_dev"] <= 1.6 * r["flips_port"] + 100 and r["rew_skipped"] < 50
    tf.assert_within_float32(r)
