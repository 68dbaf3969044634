"""The early `terminated` bytes of mir_step_begin / mir_step_go (include/mirigid.h: EARLY BYTES) in the COUPLED regime, at the
headline batch, on the launch path GenesisEnv.step uses (split step 1 = rotated launches, scene-specialised instantiation).

`terminated` is the one bit-exact requirement of the contract (reference gym_genesis/env.py:63-64), and the bytes the host gets
may precede the end of the solve they belong to.  Here that hand-over is checked where it is hardest: an arm-cube contact joins
the two trees (the bound then uses the full gradient) while the cube crosses 0.1 m inside the gripper.

  * bit for bit against a twin scene created with MIR_NO_EARLY_MASK=1 (bytes after the integrator): host masks, every device
    output, the final state;
  * against the float64 oracle's masks, free-running where the cube is clear of the threshold and teacher-forced on every step;
  * the kernel's own re-check counts no mismatch (early_mask_stats()[1] == 0), and workgroups did send early on coupled steps;
  * the production guard: a raised flag fails the next call once with MIR_E_MASK and switches the handle to late bytes.
"""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

import orc
from gym_genesis.backend import models

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOME = np.array(models.FRANKA_HOME, dtype=np.float32)
NT = max(1, min(64, len(os.sched_getaffinity(0))))
B = 4096


def _pair(monkeypatch, spec, n):
    """(scene with early bytes, twin with MIR_NO_EARLY_MASK=1), both on rotated launches"""
    from gym_genesis.backend.lib import MirScene

    monkeypatch.setenv("MIR_SPLIT_STEP", "1")
    monkeypatch.delenv("MIR_NO_EARLY_MASK", raising=False)
    sc = MirScene(spec, n)
    monkeypatch.setenv("MIR_NO_EARLY_MASK", "1")
    twin = MirScene(spec, n)
    monkeypatch.delenv("MIR_NO_EARLY_MASK", raising=False)
    assert sc.early_mask and not twin.early_mask
    return sc, twin


def _bufs(sc):
    return (sc.empty(sc.agent_dim), sc.empty(sc.env_dim), sc.empty(), sc.empty(dtype=torch.uint8))


def _grasp_workload(n, seed=5):
    """tests/golden/grasp_targets.json (4 envs x 5 stages of 40 steps) tiled to n envs, every cube moved by up to 2 mm so that no
    two envs are the same problem -> spawn positions (n, 3), actions (200, n, 9)"""
    G_ = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "grasp_targets.json")))
    T = np.array(G_["targets"], np.float32)                                   # (4, stages, 9)
    pos4 = np.array([[x, y, 0.02] for x, y in G_["cube_xy"]], np.float32)
    acts4 = np.repeat(T.transpose(1, 0, 2), G_["steps_per_stage"], axis=0)    # (200, 4, 9)
    rep = n // 4
    pos = np.tile(pos4, (rep, 1))
    pos[:, :2] += np.random.default_rng(seed).uniform(-0.002, 0.002, (n, 2)).astype(np.float32)
    return pos, np.tile(acts4, (1, rep, 1)), G_["steps_per_stage"]


def test_grasp_fixture_4096_rotated_launches_twin_and_oracle_free_running(franka_spec, monkeypatch):
    sc, twin = _pair(monkeypatch, franka_spec, B)
    assert sc.spec_active and sc.kernel == 16
    o = orc.Oracle(franka_spec, B)
    pos, acts, sps = _grasp_workload(B)
    quat = np.tile(np.array([0, 0, 0, 1], np.float32), (B, 1))
    arm = np.tile(HOME, (B, 1))
    for s in (sc, twin, o):
        s.reset(pos, quat, arm)
    b1, b2 = _bufs(sc), _bufs(twin)
    dacts = torch.as_tensor(acts, device=sc.device)
    sc.early_mask_stats(reset=True)
    sent_by_stage, last_sent = [], 0
    lifted = np.zeros(B, bool)
    compared = disagreements_near = 0
    for t in range(acts.shape[0]):
        sc.step_begin(dacts[t], *b1); h1 = sc.step_end()
        twin.step_begin(dacts[t], *b2); h2 = twin.step_end()
        assert h1.dtype == np.bool_ and np.array_equal(h1, h2), f"step {t}: early bytes differ from the twin's late bytes in {int((h1 != h2).sum())} envs"
        assert np.array_equal(h1, b1[3].cpu().numpy().astype(bool)), f"step {t}: host mask differs from the device mask"
        for x, y in zip(b1, b2):
            assert torch.equal(x, y), f"step {t}"
        o.step_batch(acts[t], NT)
        qo = o.read_all(orc.F_QPOS, o.nq)
        to = o.get_obs_all()[3].astype(bool)
        clear = np.abs(qo[:, 11] - 0.1) > 1e-3   # (free-running: away from the threshold the masks must agree bit for bit)
        assert np.array_equal(h1[clear], to[clear]), f"step {t}: {int((h1[clear] != to[clear]).sum())} masks differ from the oracle's"
        compared += int(clear.sum()); disagreements_near += int((h1[~clear] != to[~clear]).sum())
        lifted |= h1
        if (t + 1) % sps == 0:
            sent = sc.early_mask_stats()[0]
            sent_by_stage.append(sent - last_sent); last_sent = sent
    sent, bad = sc.early_mask_stats()
    print(f"\n[early mask, grasp fixture x {B}] workgroup-launches that sent early per stage {sent_by_stage} of {sps * B // 4}; mismatches {bad}; "
          f"lifted {lifted.mean():.3f}; masks compared with the oracle {compared}, within 1 mm of the threshold {200 * B - compared} "
          f"({disagreements_near} of those differ)")
    assert bad == 0
    assert lifted.mean() > 0.95
    # the close and lift stages are the coupled ones (pads on the cube): workgroups still send early there
    assert sent_by_stage[3] > 0 and sent_by_stage[4] > 0
    assert sc.early_mask   # (no MIR_E_MASK was raised)
    for x, y in zip(sc.get_state(), twin.get_state()):
        assert torch.equal(x, y)


def test_grasp_fixture_4096_teacher_forced_against_the_oracle_masks(franka_spec, monkeypatch):
    """Every one of the 200 steps from the oracle's state (rounded to float32), through step_begin / step_end with early bytes on
    (a state write makes the launch a fused one: the other instantiation that sends early)."""
    sc, _ = _pair(monkeypatch, franka_spec, B)
    o = orc.Oracle(franka_spec, B)
    pos, acts, sps = _grasp_workload(B, seed=6)
    quat = np.tile(np.array([0, 0, 0, 1], np.float32), (B, 1))
    arm = np.tile(HOME, (B, 1))
    for s in (sc, o):
        s.reset(pos, quat, arm)
    b1 = _bufs(sc)
    dacts = torch.as_tensor(acts, device=sc.device)
    sc.early_mask_stats(reset=True)
    excluded = 0
    seen_true = 0
    for t in range(acts.shape[0]):
        q, v = o.state()
        ws = o.read_all(orc.F_QACC_WS, o.nv)
        sc.set_state(qpos=q.astype(np.float32), qvel=v.astype(np.float32), warmstart=ws.astype(np.float32))
        sc.step_begin(dacts[t], *b1); h1 = sc.step_end()
        o.step_batch(acts[t], NT)
        zo = o.read_all(orc.F_QPOS, o.nq)[:, 11]
        to = o.get_obs_all()[3].astype(bool)
        clear = np.abs(zo - 0.1) > 2e-6   # (float32 against float64: one step from the same state moves z by parts in 1e7)
        excluded += int((~clear).sum())
        assert np.array_equal(h1[clear], to[clear]), f"step {t}: {int((h1[clear] != to[clear]).sum())} host masks differ from the oracle's"
        assert np.array_equal(h1, b1[3].cpu().numpy().astype(bool))
        seen_true += int(h1.sum())
    sent, bad = sc.early_mask_stats()
    print(f"\n[early mask, teacher-forced x {B}] sent early {sent} of {200 * B // 4} workgroup-launches, mismatches {bad}, env-steps within 2e-6 m of the "
          f"threshold (not compared) {excluded}, terminated env-steps {seen_true}")
    assert bad == 0 and sent > 0 and seen_true > 1000 and excluded < 50


def _example():
    spec = importlib.util.spec_from_file_location("pick_cube_state", os.path.join(ROOT, "examples", "franka", "pick_cube_state.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _expert_episode(env, ex):
    """the reference's expert (examples/franka/pick_cube_state.py) through GenesisEnv.step -> actions (T,B,9), host masks (T,B), rewards (T,B)"""
    obs, _ = env.reset(seed=0)
    acts, terms, rews = [], [], []
    for stage in ex.STAGES:
        for _ in range(40):
            a = ex.expert_policy(env.get_robot(), obs, stage)
            obs, reward, terminated, truncated, info = env.step(a)
            assert terminated.dtype == np.bool_ and not truncated.any()
            acts.append(a.clone()); terms.append(terminated.copy()); rews.append(reward.clone())
    return torch.stack(acts), np.stack(terms), torch.stack(rews).cpu().numpy()


def test_reference_expert_4096_through_genesis_env_twin_and_oracle(franka_spec, monkeypatch):
    """The reference's own expert at 4096 envs through GenesisEnv.step (the _mirfast path, rotated launches, early bytes): bit for bit
    against the same episode on an env created with MIR_NO_EARLY_MASK=1; then the recorded actions replayed on the oracle, a third
    scene teacher-forced from it on every step: host masks = the oracle's."""
    from gym_genesis.backend.lib import MirScene
    from gym_genesis.env import GenesisEnv

    ex = _example()
    monkeypatch.setenv("MIR_SPLIT_STEP", "1")
    monkeypatch.delenv("MIR_NO_EARLY_MASK", raising=False)
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
    mir = env._env._mir
    assert mir.early_mask and mir.spec_active
    mir.early_mask_stats(reset=True)
    acts, terms, rews = _expert_episode(env, ex)
    sent, bad = mir.early_mask_stats()
    monkeypatch.setenv("MIR_NO_EARLY_MASK", "1")
    env2 = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
    monkeypatch.delenv("MIR_NO_EARLY_MASK", raising=False)
    assert not env2._env._mir.early_mask
    acts2, terms2, rews2 = _expert_episode(env2, ex)
    assert torch.equal(acts, acts2) and np.array_equal(terms, terms2) and np.array_equal(rews, rews2)
    assert np.array_equal(terms, rews == 1)
    for x, y in zip(mir.get_state(), env2._env._mir.get_state()):
        assert torch.equal(x, y)
    lifted = terms.any(axis=0).mean()
    assert bad == 0 and sent > 0 and mir.early_mask
    # ---- the oracle replays the recorded actions; a scene with early bytes follows it teacher-forced
    o = orc.Oracle(franka_spec, B)
    rng = np.random.RandomState(0)   # the task's reset stream (cube_pick.py:90-91)
    pos = np.stack([rng.uniform(0.45, 0.80, B), rng.uniform(-0.25, 0.25, B), np.full(B, 0.02)], 1).astype(np.float32)
    quat = np.tile(np.array([0, 0, 0, 1], np.float32), (B, 1))
    arm = np.tile(HOME, (B, 1))
    sc = MirScene(franka_spec, B)
    assert sc.early_mask
    for s in (sc, o):
        s.reset(pos, quat, arm)
    o.step_batch(None, NT)           # reset consumes one physics step (cube_pick.py:107)
    b1 = _bufs(sc)
    hacts = acts.cpu().numpy()
    excluded = 0
    for t in range(hacts.shape[0]):
        q, v = o.state()
        ws = o.read_all(orc.F_QACC_WS, o.nv)
        sc.set_state(qpos=q.astype(np.float32), qvel=v.astype(np.float32), warmstart=ws.astype(np.float32))
        sc.step_begin(acts[t], *b1); h1 = sc.step_end()
        o.step_batch(hacts[t], NT)
        zo = o.read_all(orc.F_QPOS, o.nq)[:, 11]
        to = o.get_obs_all()[3].astype(bool)
        clear = np.abs(zo - 0.1) > 2e-6
        excluded += int((~clear).sum())
        assert np.array_equal(h1[clear], to[clear]), f"step {t}: {int((h1[clear] != to[clear]).sum())} host masks differ from the oracle's"
    sent2, bad2 = sc.early_mask_stats()
    print(f"\n[early mask, reference expert x {B}] GenesisEnv.step: sent early {sent} of {200 * B // 4} workgroup-launches, mismatches {bad}, lifted "
          f"{lifted:.3f}; teacher-forced replay: sent {sent2}, mismatches {bad2}, env-steps not compared {excluded}")
    assert bad2 == 0 and excluded < 50


def test_solver_stopped_by_its_iteration_cap_still_sends_right_bytes(monkeypatch):
    """The bound holds however the solver stops: with the cap at 2 Newton iterations most contact-rich envs leave the loop
    unconverged.  Cubes thrown through the threshold, the scripted grasp: early bytes = late bytes, no mismatch counted."""
    n = 1024
    sb = models.franka_cube_pick_scene()
    sb.opt["iterations"] = 2
    spec = sb.build()
    sc, twin = _pair(monkeypatch, spec, n)
    assert not sc.spec_active
    pos, acts, sps = _grasp_workload(n, seed=8)
    quat = np.tile(np.array([0, 0, 0, 1], np.float32), (n, 1))
    arm = np.tile(HOME, (n, 1))
    b1, b2 = _bufs(sc), _bufs(twin)
    dacts = torch.as_tensor(acts, device=sc.device)
    g = np.random.default_rng(3)
    racts = torch.as_tensor(g.uniform(-1, 1, (64, n, 9)).astype(np.float32), device=sc.device)
    sc.early_mask_stats(reset=True)
    sc.set_diag(True)
    capped = total = trues = 0
    for phase in range(3):
        for s in (sc, twin):
            s.reset(pos, quat, arm)
        if phase > 0:   # cubes thrown upwards through the threshold (and falling back through it) under random arm targets
            q, v, tg, ws = sc.get_state()
            v2 = v.clone(); v2[::2, 11] = 2.0 + phase
            for s in (sc, twin):
                s.set_state(qpos=q, qvel=v2, target=tg, warmstart=ws)
        for t in range(200 if phase == 0 else 80):
            a = dacts[t] if phase == 0 else racts[(t + 7 * phase) % 64]
            sc.step_begin(a, *b1); h1 = sc.step_end()
            twin.step_begin(a, *b2); h2 = twin.step_end()
            assert np.array_equal(h1, h2), f"phase {phase} step {t}"
            for x, y in zip(b1, b2):
                assert torch.equal(x, y)
            it = sc.get_diag()[2].cpu().numpy()
            capped += int((it >= 2).sum()); total += n; trues += int(h1.sum())
    sent, bad = sc.early_mask_stats()
    print(f"\n[early mask, iteration cap 2] env-steps at the cap {capped} of {total}; sent early {sent}; mismatches {bad}; terminated env-steps {trues}")
    assert bad == 0 and sent > 0 and capped > 0.05 * total and trues > 100


def test_mask_flag_fails_the_next_call_once_and_switches_to_late_bytes(franka_spec, monkeypatch):
    """The production guard (MIR_E_MASK): what a kernel that found its early bytes wrong does -- raise the sticky word -- done by hand."""
    from gym_genesis.backend.lib import MirMaskError
    from gym_genesis.env import GenesisEnv

    n = 256
    sc, twin = _pair(monkeypatch, franka_spec, n)
    rng = np.random.RandomState(0)
    pos = np.stack([rng.uniform(0.45, 0.80, n), rng.uniform(-0.25, 0.25, n), np.full(n, 0.02)], 1).astype(np.float32)
    quat = np.tile(np.array([0, 0, 0, 1], np.float32), (n, 1))
    arm = np.tile(HOME, (n, 1))
    for s in (sc, twin):
        s.reset(pos, quat, arm)
    b1, b2 = _bufs(sc), _bufs(twin)
    acts = torch.as_tensor(np.random.default_rng(0).uniform(-1, 1, (12, n, 9)).astype(np.float32), device=sc.device)
    for t in range(4):
        sc.step_begin(acts[t], *b1); sc.step_end()
        twin.step_begin(acts[t], *b2); twin.step_end()
    assert sc.lib.mir_debug_raise_mask_flag(sc.h) == 0
    with pytest.raises(MirMaskError):
        sc.step_begin(acts[4], *b1)
    assert not sc.early_mask          # late bytes from here on
    for t in range(4, 8):             # ... and the scene goes on, in step with the twin (the failed call launched nothing)
        sc.step_begin(acts[t], *b1); h1 = sc.step_end()
        twin.step_begin(acts[t], *b2); h2 = twin.step_end()
        assert np.array_equal(h1, h2)
        for x, y in zip(b1, b2):
            assert torch.equal(x, y)
    sent_before = sc.early_mask_stats()[0]
    sc.step_begin(acts[8], *b1); sc.step_end()
    assert sc.early_mask_stats()[0] == sent_before   # nothing is sent early any more
    # raised between begin and end: mir_step_end reports it, the step stays open and the next call closes it
    sc2, _ = _pair(monkeypatch, franka_spec, n)
    sc2.reset(pos, quat, arm)
    sc2.step_begin(acts[0], *b1)
    assert sc2.lib.mir_debug_raise_mask_flag(sc2.h) == 0
    with pytest.raises(MirMaskError):
        sc2.step_end()
    sc2.step_begin(acts[1], *b1); sc2.step_end()
    # mir_reset checks it too
    assert sc2.early_mask is False
    sc3, _ = _pair(monkeypatch, franka_spec, n)
    assert sc3.lib.mir_debug_raise_mask_flag(sc3.h) == 0
    with pytest.raises(MirMaskError):
        sc3.reset(pos, quat, arm)
    sc3.reset(pos, quat, arm)
    # through GenesisEnv.step (the _mirfast path): raises once, then the env goes on
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=n, enable_pixels=False)
    env.reset(seed=0)
    env.step(acts[0])
    mir = env._env._mir
    assert mir.lib.mir_debug_raise_mask_flag(mir.h) == 0
    with pytest.raises(MirMaskError):
        env.step(acts[1])
    obs, reward, terminated, truncated, info = env.step(acts[2])
    assert terminated.shape == (n,) and not mir.early_mask
