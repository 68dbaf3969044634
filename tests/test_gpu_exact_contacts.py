"""EXACT CONTACTS (include/mirigid.h: mir_set_exact_contacts; GenesisEnv(..., exact_contacts=True)).

Genesis keeps every contact point of its candidate pairs (RigidOptions at /root/reference/gym_genesis/tasks/franka/cube_pick.py:46);
the 16-lane kernel keeps 16 per env and thins the manifolds beyond that -- in 29 % of the env-steps of the reference's own expert
(/root/reference/examples/franka/pick_cube_state.py:33-41,86-88).  With the switch on (the default of the pick tasks since round 6), a
step DEFERS exactly the envs whose narrowphase found more than 16 points; mir_step_end steps those with 48 points, never thinned here --
on the LIST INSTANTIATION of the 16-lane kernel (three contacts per lane; round 5: on the wave-per-env kernel, which is still where an
env with more than 16 candidate pairs goes, and every deferred env under MIR_EXACT_WAVE=1) -- which also writes their half of the split
step's hand-over.  What is checked:

  * the reference's expert at 4096 envs: teacher-forced on all 200 steps against the float64 oracle AT CAPACITY 48 -- joint state
    of every env, deferred ones included, and the host masks bit for bit;
  * free-running on the rotated launches (the path GenesisEnv.step takes): on every step every env is bit-identical to one of two
    teacher-forced twins -- the plain 16-lane scene where the env was not deferred (no thinning there: the same computation), the
    scene whose EVERY env takes the deferred envs' route (set_exact_contacts("all")) where it was; and that twin stays within float32
    rounding of the same scene on the wave-per-env kernel (contact_capacity = 48), last round's twin;
  * a workload without overflow: switch on == switch off, bit for bit, nothing extra launched;
  * the entry points that cannot close their steps on the host are refused, the others wait.
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


def _spec48():
    sb = models.franka_cube_pick_scene()
    sb.opt["max_contacts"] = 48
    return sb.build()


def _bufs(sc):
    return (sc.empty(sc.agent_dim), sc.empty(sc.env_dim), sc.empty(), sc.empty(dtype=torch.uint8))


def _example():
    spec = importlib.util.spec_from_file_location("pick_cube_state", os.path.join(ROOT, "examples", "franka", "pick_cube_state.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _grasp_workload(n, seed=5):
    """tests/golden/grasp_targets.json tiled to n envs, cubes moved by up to 2 mm (as tests/test_gpu_early_mask.py)"""
    G_ = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "grasp_targets.json")))
    T = np.array(G_["targets"], np.float32)
    pos4 = np.array([[x, y, 0.02] for x, y in G_["cube_xy"]], np.float32)
    acts4 = np.repeat(T.transpose(1, 0, 2), G_["steps_per_stage"], axis=0)
    rep = n // 4
    pos = np.tile(pos4, (rep, 1))
    pos[:, :2] += np.random.default_rng(seed).uniform(-0.002, 0.002, (n, 2)).astype(np.float32)
    return pos, np.tile(acts4, (1, rep, 1))


def test_reference_expert_4096_exact_contacts_teacher_forced_against_the_capacity_48_oracle(franka_spec, monkeypatch):
    """The expert's actions come from an episode through GenesisEnv(exact_contacts=True).step (rotated launches, deferred envs on the
    wave kernel); the oracle at capacity 48 replays them, and a scene with the switch on follows it teacher-forced."""
    from gym_genesis.backend.lib import MirScene
    from gym_genesis.env import GenesisEnv

    ex = _example()
    monkeypatch.setenv("MIR_SPLIT_STEP", "1")
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False, exact_contacts=True)
    mir = env._env._mir
    assert mir.kernel == 16 and mir.exact_contacts and mir.split_step == 1
    obs, _ = env.reset(seed=0)
    mir.exact_stats(reset=True)
    acts, terms = [], []
    for stage in ex.STAGES:
        for _ in range(40):
            a = ex.expert_policy(env.get_robot(), obs, stage)
            obs, reward, terminated, truncated, info = env.step(a)
            assert terminated.dtype == np.bool_ and np.array_equal(terminated, (reward == 1).cpu().numpy())
            assert np.array_equal(terminated, info["is_success"].cpu().numpy())
            acts.append(a.clone()); terms.append(terminated.copy())
    st_env = mir.exact_stats()
    lifted = np.stack(terms).any(0).mean()
    assert st_env["steps"] == 200 and st_env["overflow_env_steps"] > 0.1 * 200 * B and st_env["overflow_envs_max"] <= B
    assert torch.isfinite(obs["environment_state"]).all() and lifted > 0.5
    # ---- the capacity-48 oracle replays the actions; a scene with the switch on follows it from the oracle's state on every step, and
    # so does the float32 CPU port of the oracle at the same capacity (the yardstick for what float32 arithmetic gives on this workload)
    spec48 = _spec48()
    o, port = orc.Oracle(spec48, B), orc.Oracle(spec48, B, f32="big")
    rng = np.random.RandomState(0)   # the task's reset stream (cube_pick.py:90-91)
    pos = np.stack([rng.uniform(0.45, 0.80, B), rng.uniform(-0.25, 0.25, B), np.full(B, 0.02)], 1).astype(np.float32)
    quat = np.tile(np.array([0, 0, 0, 1], np.float32), (B, 1))
    arm = np.tile(HOME, (B, 1))
    sc = MirScene(franka_spec, B)
    sc.set_exact_contacts(True)
    sc.set_diag(True)
    for s in (sc, o, port):
        s.reset(pos, quat, arm)
    o.step_batch(None, NT)           # reset consumes one physics step (cube_pick.py:107)
    b1 = _bufs(sc)
    sc.exact_stats(reset=True)
    e_dev, e_port, was_def = [], [], []
    excluded = flips = deferred = cls_flips = 0
    for t in range(200):
        q, v = o.state()
        ws, tg = o.read_all(orc.F_QACC_WS, o.nv), o.read_all(orc.F_TARGET, o.nv)
        q32, v32, w32 = q.astype(np.float32), v.astype(np.float32), ws.astype(np.float32)
        sc.set_state(qpos=q32, qvel=v32, warmstart=w32)
        for f, x in ((orc.F_QPOS, q32), (orc.F_QVEL, v32), (orc.F_QACC_WS, w32), (orc.F_TARGET, tg)):
            port.write_all(f, x)
        sc.step_begin(acts[t], *b1); h1 = sc.step_end()
        a = acts[t].cpu().numpy()
        o.step_batch(a, NT)
        port.step_batch(a, NT)
        qo = o.state()[0]
        to = o.get_obs_all()[3].astype(bool)
        npts = o.ncand_all()
        ncon_dev, _, _, pts_dev = (x.cpu().numpy() for x in sc.get_diag(points=True))
        # (the device defers the envs whose candidate points exceed 16 by ITS count; a point exactly at make / break may exist on one
        #  side only after the float32 rounding of the injected state: such env-steps are counted, excluded below, and must be rare)
        deferred += int((pts_dev > 16).sum())
        cls_flips += int(((pts_dev > 16) != (npts > 16)).sum())
        clear = np.abs(qo[:, 11] - 0.1) > 2e-6
        excluded += int((~clear).sum())
        assert np.array_equal(h1[clear], to[clear]), f"step {t}: {int((h1[clear] != to[clear]).sum())} host masks differ from the oracle's"
        assert np.array_equal(h1, b1[3].cpu().numpy().astype(bool))
        qh = sc.get_state()[0].cpu().numpy()
        same = (ncon_dev == o.counts_all()[0]) & (ncon_dev == port.counts_all()[0])   # (contact-count flips: as above)
        flips += int((~same).sum())
        e_dev.append(np.abs(qh - qo).max(1)[same]); e_port.append(np.abs(port.state()[0] - qo).max(1)[same]); was_def.append((pts_dev > 16)[same])
    st = sc.exact_stats()
    e_dev, e_port, was_def = np.concatenate(e_dev), np.concatenate(e_port), np.concatenate(was_def)
    qs = (0.5, 0.9, 0.99, 0.999, 0.9999)
    fmt = lambda x: " ".join(f"{np.quantile(x, q):.1e}" for q in qs) + f" max {x.max():.1e}"  # noqa: E731
    print(f"\n[exact contacts, reference expert x {B}] GenesisEnv.step: deferred env-steps {st_env['overflow_env_steps']} of {200 * B} in {st_env['overflow_steps']} "
          f"steps (most in one step {st_env['overflow_envs_max']}), lifted {lifted:.3f}.  Teacher-forced against the capacity-48 float64 oracle, one-step qpos "
          f"L-inf, quantiles {qs}:\n  deferred envs (list instantiation)  device {fmt(e_dev[was_def])} | float32 CPU port {fmt(e_port[was_def])}\n  other envs (16-lane kernel) "
          f" device {fmt(e_dev[~was_def])} | float32 CPU port {fmt(e_port[~was_def])}\n  deferred env-steps {st['overflow_env_steps']} (the oracle's count of envs "
          f"with more than 16 points differs in {cls_flips}); masks not compared (within 2e-6 m of the threshold) {excluded}; contact-count flips excluded {flips}")
    # (an env is deferred on the 16-lane kernel's count; `deferred` is read back from the diagnostics, which the WAVE kernel wrote for
    #  a deferred env -- its own count of the same state, and at make / break the two narrowphases may differ by a point)
    assert abs(st["overflow_env_steps"] - deferred) <= 20 and deferred > 0.1 * 200 * B
    assert excluded < 50 and flips < 200 * B // 500 and cls_flips <= flips
    # the bar: no further from float64 than the float32 CPU port of the oracle is, quantile by quantile (factor 1.5 + 2e-6 as in
    # tests/test_gpu_parity.py) -- on the deferred envs (up to 35 contact points on the wave kernel) and on the others alike
    for sel in (was_def, ~was_def):
        for qn in qs:
            assert np.quantile(e_dev[sel], qn) <= 1.5 * np.quantile(e_port[sel], qn) + 2e-6, (qn, np.quantile(e_dev[sel], qn), np.quantile(e_port[sel], qn))
    assert np.quantile(e_dev, 0.999) < 5e-6


def _free_running_against_twins(franka_spec, n, want_split):
    """Scripted grasp at n envs, switch on, free-running.  Before every step the state goes to two twins (a state write: fused launches
    there): the plain 16-lane scene, and the scene whose every env takes the route of the deferred ones -- the list instantiation of the
    16-lane kernel (under MIR_EXACT_WAVE=1: the pick scene on the wave kernel, contact_capacity = 48).  After the step every env's
    outputs and state rows equal the plain twin's where it was not deferred and the other twin's where it was.  A third scene, the pick
    scene on the wave kernel, says how far the list instantiation is from last round's route (float32 rounding).  -> counters"""
    from gym_genesis.backend.lib import MirScene

    sc = MirScene(franka_spec, n)
    sc.set_exact_contacts(True)
    via_wave = os.environ.get("MIR_EXACT_WAVE", "0") not in ("", "0")
    plain, wave = MirScene(franka_spec, n), MirScene(_spec48(), n)
    if via_wave:
        twin = wave
    else:
        twin = MirScene(franka_spec, n)
        twin.set_exact_contacts("all")
    assert sc.kernel == 16 and plain.kernel == 16 and wave.kernel == 64 and sc.split_step == want_split
    for s in (sc, plain, wave, twin):
        s.set_diag(True)
    pos, acts = _grasp_workload(n)
    quat = np.tile(np.array([0, 0, 0, 1], np.float32), (n, 1))
    arm = np.tile(HOME, (n, 1))
    sc.reset(pos, quat, arm)
    b0, b1, b2, b3 = _bufs(sc), _bufs(plain), _bufs(twin), _bufs(wave)
    dacts = torch.as_tensor(acts, device=sc.device)
    sc.exact_stats(reset=True)
    n_def = steps_def = 0
    lifted = np.zeros(n, bool)
    far = []
    for t in range(acts.shape[0]):
        st = sc.get_state()
        for tw in {plain, twin, wave}:
            tw.set_state(*st)
        sc.step_begin(dacts[t], *b0); h0 = sc.step_end()
        plain.step_fused(dacts[t], *b1)
        twin.step_fused(dacts[t], *b2)
        if twin is not wave:
            wave.step_fused(dacts[t], *b3)
        dfr = sc.get_diag(points=True)[3] > 16           # (a deferred env's record is the wave kernel's: its count of the same state)
        # (the 16-lane kernel defers on ITS count; at make / break the two narrowphases may differ by a point: such an env is in
        #  neither class by the counts alone -- it is recognised by which twin it equals, below)
        n_def += int(dfr.sum()); steps_def += int(dfr.any())
        s0, s1, s2 = sc.get_state(), plain.get_state(), twin.get_state()
        if twin is not wave and bool(dfr.any()):
            far.append((s2[0] - wave.get_state()[0]).abs().max(1).values[dfr].cpu().numpy())
        eq_plain = torch.ones(n, dtype=torch.bool, device=sc.device)
        eq_wave = torch.ones(n, dtype=torch.bool, device=sc.device)
        for x, y, z in zip(list(b0) + list(s0), list(b1) + list(s1), list(b2) + list(s2)):
            eq_plain &= (x == y).reshape(n, -1).all(1)
            eq_wave &= (x == z).reshape(n, -1).all(1)
        assert bool((eq_plain | eq_wave).all()), f"step {t}: {int((~(eq_plain | eq_wave)).sum())} envs equal neither twin"
        assert bool(eq_plain[~dfr].all()) or int((~eq_plain[~dfr]).sum()) <= 2, f"step {t}: envs that were not deferred differ from the plain 16-lane scene"
        assert bool(eq_wave[dfr].all()), f"step {t}: a deferred env differs from the scene whose every env takes the deferred envs' route"
        assert np.array_equal(h0, b0[3].cpu().numpy().astype(bool))
        lifted |= h0
    route = sc.exact_route()
    if via_wave:
        assert route["list_env_steps"] == 0 and route["wave_env_steps"] == sc.exact_stats()["overflow_env_steps"]
    else:
        # (this workload stays under 48 points and 16 candidate pairs: nothing reaches the wave-per-env kernel; in a HEAVY phase -- at
        #  least 1 / 16 of the envs above 16 points -- the whole batch takes one launch of the three-contacts-per-lane instantiation and
        #  there is no list)
        #  (an OVERFLOW RUN -- the default on the rotated launches since the second session of round 6: from the step after the first
        #   deferral the whole batch takes the rotated launch with three contacts per lane, and only the envs whose 48-point scratch row
        #   the launch before could not write go through the list)
        assert route["list_env_steps"] <= sc.exact_stats()["overflow_env_steps"] and route["wave_env_steps"] == 0
        assert route["list_env_steps"] == sc.exact_stats()["overflow_env_steps"] or route["heavy_steps"] > 0 or route["big_steps"] > 0
        if want_split == 1 and not any(os.environ.get(k) for k in ("MIR_EXACT_ONE_STREAM", "MIR_EXACT_WAVE")) and os.environ.get("MIR_EXACT_BIG", "1") != "0":
            assert route["big_steps"] > 0 and route["heavy_steps"] == 0, route   # (this loop spends milliseconds between two steps)
        far = np.concatenate(far)
        print(f"\n[list instantiation against the wave-per-env kernel, same state, same action, one step, {far.size} deferred env-steps] qpos L-inf "
              f"median {np.median(far):.1e} 0.99 {np.quantile(far, 0.99):.1e} 0.9999 {np.quantile(far, 0.9999):.1e} max {far.max():.1e}")
        assert np.quantile(far, 0.99) < 5e-6 and np.quantile(far, 0.9999) < 2e-4
    return sc.exact_stats(), n_def, steps_def, lifted


def test_free_running_rotated_launches_every_env_equals_its_twin_bit_for_bit(franka_spec, monkeypatch):
    """4096 envs on the rotated launches (the path GenesisEnv.step takes), deferred envs' launches on the side stream."""
    monkeypatch.setenv("MIR_SPLIT_STEP", "1")
    st, n_def, steps_def, lifted = _free_running_against_twins(franka_spec, B, 1)
    print(f"\n[exact contacts, grasp fixture x {B}, rotated launches] deferred env-steps {n_def} in {steps_def} of 200 steps (most in one step "
          f"{st['overflow_envs_max']}); lifted {lifted.mean():.3f}")
    assert abs(st["overflow_env_steps"] - n_def) <= 20 and st["overflow_steps"] >= steps_def and n_def > 1000 and st["steps"] == 200
    assert lifted.mean() > 0.9


@pytest.mark.parametrize("var,val,split", [("MIR_SPLIT_STEP", "0", 0), ("MIR_SPLIT_STEP", "2", 2), ("MIR_NO_EARLY_MASK", "1", 1), ("MIR_EXACT_ONE_STREAM", "1", 1),
                                           ("MIR_EXACT_WAVE", "1", 1), ("MIR_EXACT_BIG", "0", 1)])
def test_every_launch_kind_defers_the_same_way(franka_spec, monkeypatch, var, val, split):
    """The other ways a step is launched -- one fused launch per step, the split step as two launches, terminated bytes that wait for
    the integrator, the deferred envs' launches on the step's own stream, the deferred envs on the wave-per-env kernel (round 5's route,
    the fallback of this round's), list launches and heavy phase even in a loop that leaves room between its steps (MIR_EXACT_BIG=0; the
    default takes the two-launch steps of an overflow run in these loops) -- at 512 envs: every env of every step equals its twin."""
    monkeypatch.setenv(var, val)
    st, n_def, steps_def, lifted = _free_running_against_twins(franka_spec, 512, split)
    assert abs(st["overflow_env_steps"] - n_def) <= 10 and n_def > 100 and st["steps"] == 200 and lifted.mean() > 0.9


@pytest.mark.parametrize("route", ["heavy", "two_launches"])
def test_heavy_phase_from_the_first_overflow_on_and_a_batch_that_is_not_a_multiple_of_four(franka_spec, monkeypatch, route):
    """MIR_EXACT_HEAVY=1,1: the whole batch goes to the three-contacts-per-lane launch from the step after the first deferred env until no
    env is above 16 points -- with the envs served in sorted order (the ones above 16 points first), 30 envs: the last workgroup holds
    two.  Every env of every step equals its twin (an env with at most 16 points is the one-contact-per-lane kernel's bit for bit)."""
    from gym_genesis.backend.lib import MirScene

    monkeypatch.setenv("MIR_SPLIT_STEP", "1")
    if route == "heavy":
        monkeypatch.setenv("MIR_EXACT_HEAVY", "1,1")
        monkeypatch.setenv("MIR_EXACT_BIG", "0")   # (this loop spends milliseconds between two steps: it would otherwise take the two-launch steps of an overflow run)
    else:   # (the same 30 envs through the overflow runs' two launches, their second half as two lists padded to whole workgroups)
        monkeypatch.setenv("MIR_EXACT_BIG", "2")
    n = 30
    sc, plain, twin = MirScene(franka_spec, n), MirScene(franka_spec, n), MirScene(franka_spec, n)
    sc.set_exact_contacts(True)
    twin.set_exact_contacts("all")
    for s_ in (sc, plain, twin):
        s_.set_diag(True)
    pos, acts = _grasp_workload(32)
    pos, acts = pos[:n], acts[:, :n]
    sc.reset(pos, np.tile(np.array([0, 0, 0, 1], np.float32), (n, 1)), np.tile(HOME, (n, 1)))
    b0, b1, b2 = _bufs(sc), _bufs(plain), _bufs(twin)
    dacts = torch.as_tensor(acts, device=sc.device)
    sc.exact_stats(reset=True)
    n_def = 0
    for t in range(acts.shape[0]):
        st = sc.get_state()
        for tw in (plain, twin):
            tw.set_state(*st)
        sc.step_begin(dacts[t], *b0); h0 = sc.step_end()
        plain.step_fused(dacts[t], *b1)
        twin.step_fused(dacts[t], *b2)
        dfr = sc.get_diag(points=True)[3] > 16
        n_def += int(dfr.sum())
        for x, y, z in zip(list(b0) + list(sc.get_state()), list(b1) + list(plain.get_state()), list(b2) + list(twin.get_state())):
            assert torch.equal(x[~dfr], y[~dfr]), f"step {t}: an env with at most 16 points differs from the plain scene"
            assert torch.equal(x[dfr], z[dfr]), f"step {t}: an env above 16 points differs from its twin"
        assert np.array_equal(h0, b0[3].cpu().numpy().astype(bool))
    st, r = sc.exact_stats(), sc.exact_route()
    assert n_def > 20 and st["overflow_env_steps"] == n_def and r["wave_env_steps"] == 0 and (r["heavy_steps"] > 3 if route == "heavy" else r["big_steps"] > 3), (n_def, st, r)


@pytest.mark.parametrize("big", [False, True])
def test_so101_scene_with_a_low_capacity_random_actions_every_env_equals_its_twin_bit_for_bit(monkeypatch, big):
    """The other articulation (BASELINE configs[3]) and another way to overflow: the SO-101 pick scene with the capacity of the
    16-lane kernel set to 4 points -- the cube resting on the slab -- and random joint targets: whenever the arm touches the slab or the
    cube the env overflows.  Switch on, rotated launches.  Every env of every step equals the plain scene (capacity 4) where it was not
    deferred and the same scene with every env on the deferred envs' route (the list instantiation, capacity 48) where it was -- envs
    with 5 .. 16 points included, which the 16-lane kernel could have held."""
    from gym_genesis.backend.lib import MirScene

    def spec(cap):
        sb = models.so101_cube_pick_scene()
        sb.opt["max_contacts"] = cap
        return sb.build()

    n = 2048
    monkeypatch.setenv("MIR_SPLIT_STEP", "1")
    monkeypatch.setenv("MIR_EXACT_BIG", "2" if big else "0")   # (the overflow steps as two launches for the whole batch: the generic-scene instantiations of them)
    sc, plain, wave = MirScene(spec(4), n), MirScene(spec(4), n), MirScene(spec(4), n)
    sc.set_exact_contacts(True)
    wave.set_exact_contacts("all")
    assert sc.kernel == 16 and plain.kernel == 16 and wave.kernel == 16
    for s_ in (sc, plain, wave):
        s_.set_diag(True)
    rng = np.random.RandomState(4)
    pos = np.stack([rng.uniform(-0.32, -0.28, n), rng.uniform(-0.05, 0.05, n), np.full(n, models.ISLAND_TOP_Z + 0.021)], 1).astype(np.float32)
    sc.reset(pos, np.tile(np.array([1, 0, 0, 0], np.float32), (n, 1)), np.zeros((n, 6), np.float32))
    b0, b1, b2 = _bufs(sc), _bufs(plain), _bufs(wave)
    acts = torch.as_tensor(np.random.default_rng(8).uniform(-1.6, 1.6, (200, n, 6)).astype(np.float32), device=sc.device)
    sc.exact_stats(reset=True)
    n_def = 0
    for t in range(acts.shape[0]):
        st = sc.get_state()
        for tw in (plain, wave):
            tw.set_state(*st)
        sc.step_begin(acts[t], *b0); h0 = sc.step_end()
        plain.step_fused(acts[t], *b1)
        wave.step_fused(acts[t], *b2)
        dfr = sc.get_diag(points=True)[3] > 4
        n_def += int(dfr.sum())
        for x, y, z in zip(list(b0) + list(sc.get_state()), list(b1) + list(plain.get_state()), list(b2) + list(wave.get_state())):
            assert torch.equal(x[~dfr], y[~dfr]), f"step {t}: an env that was not deferred differs from the plain scene"
            assert torch.equal(x[dfr], z[dfr]), f"step {t}: a deferred env differs from the scene whose every env takes the deferred envs' route"
        assert np.array_equal(h0, b0[3].cpu().numpy().astype(bool))
    st = sc.exact_stats()
    r = sc.exact_route()
    assert r["wave_env_steps"] == 0 and r["heavy_steps"] == 0 and 0 < r["list_env_steps"] <= st["overflow_env_steps"], (r, st)
    assert (r["big_steps"] > 0 and r["list_env_steps"] < st["overflow_env_steps"]) if big else (r["big_steps"] == 0 and r["list_env_steps"] == st["overflow_env_steps"]), (r, st)
    print(f"\n[exact contacts, SO-101 at capacity 4 x {n}, random targets] deferred env-steps {n_def} of {200 * n} in {st['overflow_steps']} of 200 steps")
    assert abs(st["overflow_env_steps"] - n_def) <= 20 and n_def > 300


def test_without_overflow_the_switch_changes_nothing(franka_spec, monkeypatch):
    """The headline workload (random targets, cube at rest): no env ever has more than 16 candidate points; switch on == switch off
    bit for bit through GenesisEnv.step, and no step launched anything extra."""
    from gym_genesis.env import GenesisEnv

    a = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False, exact_contacts=True)
    b = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False, exact_contacts=False)
    assert a._env._mir.exact_contacts and not b._env._mir.exact_contacts
    assert GenesisEnv(task="cube_pick", robot="franka", num_envs=4, enable_pixels=False)._env._mir.exact_contacts   # (the default since round 6)
    a.reset(seed=1); b.reset(seed=1)
    a._env._mir.exact_stats(reset=True)
    acts = torch.as_tensor(np.random.default_rng(2).uniform(-1, 1, (60, B, 9)).astype(np.float32), device=a._env.device)
    for t in range(60):
        oa, ra, ta, _, _ = a.step(acts[t])
        ob, rb, tb, _, _ = b.step(acts[t])
        assert np.array_equal(ta, tb) and torch.equal(ra, rb)
        assert torch.equal(oa["agent_pos"], ob["agent_pos"]) and torch.equal(oa["environment_state"], ob["environment_state"])
    st = a._env._mir.exact_stats()
    assert st == {"steps": 60, "overflow_steps": 0, "overflow_env_steps": 0, "overflow_envs_max": 0}
    for x, y in zip(a._env._mir.get_state(), b._env._mir.get_state()):
        assert torch.equal(x, y)


def test_entry_points_under_the_switch(franka_spec):
    from gym_genesis.backend.lib import MirError, MirScene

    n = 64
    sc = MirScene(franka_spec, n)
    sc.set_exact_contacts(True)
    rng = np.random.RandomState(0)
    pos = np.stack([rng.uniform(0.45, 0.80, n), rng.uniform(-0.25, 0.25, n), np.full(n, 0.02)], 1).astype(np.float32)
    sc.reset(pos, np.tile(np.array([0, 0, 0, 1], np.float32), (n, 1)), np.tile(HOME, (n, 1)))
    sc.step(3)                                                   # begin + end per step
    a = sc.get_state()[2].clone()
    sc.step_fused(a, *_bufs(sc))                                 # begin + end
    assert sc.exact_stats()["steps"] == 4
    rows = torch.zeros((2, n, 24), device=sc.device)
    with pytest.raises(MirError):
        sc.step_packed(a, rows[0])
    with pytest.raises(MirError):
        sc.rollout(a.repeat(2, 1, 1).contiguous(), rows)
    sc.set_exact_contacts(False)
    sc.rollout(a.repeat(2, 1, 1).contiguous(), rows)            # (and back off: everything is available again)
    sc.set_exact_contacts(True)
    sc.step(1)
    # (needs the tagged terminated bytes: refused in the other sync modes)
    import os
    os.environ["MIR_SYNC_MODE"] = "0"
    try:
        s0 = MirScene(franka_spec, n)
        with pytest.raises(MirError):
            s0.set_exact_contacts(True)
    finally:
        del os.environ["MIR_SYNC_MODE"]
    # a scene that already runs on the wave kernel has nothing to switch
    w = MirScene(_spec48(), n)
    assert w.kernel == 64
    with pytest.raises(MirError):
        w.set_exact_contacts(True)


def test_pixels_behind_a_step_with_deferred_envs_show_the_stepped_state():
    """enable_pixels with exact contacts: the images of the observation are drawn behind the step -- for a deferred env behind the wave
    kernel's launch on the side stream, whose link poses the 16-lane launch did not write (the render refreshes them).  The pixels of
    every env equal a fresh render of the state the step left, and the state equals the state-only env's."""
    from gym_genesis.env import GenesisEnv

    n = 64
    kw = dict(task="cube_pick", robot="franka", num_envs=n, exact_contacts=True)
    a = GenesisEnv(enable_pixels=True, observation_height=96, observation_width=128, camera_capture_mode="per_env", strip_environment_state=False, **kw)
    b = GenesisEnv(enable_pixels=False, **kw)
    a.reset(seed=3); b.reset(seed=3)
    pos, acts = _grasp_workload(n, seed=9)
    quat = torch.tensor([[0.0, 0.0, 0.0, 1.0]]).repeat(n, 1)
    home = torch.tensor(HOME).repeat(n, 1)
    for e in (a, b):
        e._env._mir.reset(pos, quat, home)
        e._env._mir.exact_stats(reset=True)
    deferred_frames = 0
    for t in range(100, 140):   # (the close stage: pads on the cube, fingertips on the floor)
        act = torch.as_tensor(acts[t], device=a._env.device)
        oa, ra, ta, _, _ = a.step(act)
        ob, rb, tb, _, _ = b.step(act)
        assert np.array_equal(ta, tb) and torch.equal(oa["agent_pos"], ob["agent_pos"])
        fresh = a._env.cam.render_envs()
        assert torch.equal(oa["pixels"], fresh), f"step {t}: the observation's images are not those of the stepped state"
        deferred_frames = a._env._mir.exact_stats()["overflow_env_steps"]
    assert deferred_frames > 0 and a._env._mir.exact_stats() == b._env._mir.exact_stats()


def test_more_candidate_pairs_than_lanes_defers_too():
    """The 16-lane kernel's other capacity: 16 candidate PAIRS (lane = candidate); what passes the broadphase beyond that is dropped.
    Two rigid combs of eleven small boxes, one lying on the floor, the other resting on it tooth on tooth: 22 touching pairs.  With exact
    contacts such an env reports its candidate-point count saturated and goes to the wave kernel (64 candidates, 48 points): every step
    of every env equals the same scene on the wave kernel bit for bit."""
    from gym_genesis.backend import spec as S
    from gym_genesis.backend.lib import MirScene

    def comb(cap):
        sb = S.SceneBuilder()
        sb.add_geom(0, S.GEOM_PLANE)
        for name, z in (("a", 0.02), ("b", 0.0595)):
            sb.add_body(name, 0, pos=(0.0, 0.0, z), jtype=S.JNT_FREE, mass=0.55, inertia=S.box_inertia(0.55, (0.27, 0.02, 0.02)))
            for i in range(11):
                sb.add_geom(name, S.GEOM_BOX, size=(0.02, 0.02, 0.02), pos=(0.05 * (i - 5), 0.0, 0.0))
        sb.task = dict(eef_body=1, obj_body=2, grip_dof=(), reward_z=0.1)
        sb.opt["max_contacts"] = cap
        return sb.build()

    n = 64
    sc, wave = MirScene(comb(16), n), MirScene(comb(48), n)
    assert sc.kernel == 16 and wave.kernel == 64
    sc.set_exact_contacts(True)
    sc.set_diag(True)
    rng = np.random.default_rng(2)
    q = np.zeros((n, 14), np.float32)
    q[:, 2], q[:, 9] = 0.0199, 0.0595
    q[:, 7:9] = rng.uniform(-0.003, 0.003, (n, 2))
    q[:, 3], q[:, 10] = 1.0, 1.0
    for s_ in (sc, wave):
        s_.set_state(qpos=q, qvel=np.zeros((n, 12), np.float32), warmstart=np.zeros((n, 12), np.float32))
    b0, b2 = _bufs(sc), _bufs(wave)
    sc.exact_stats(reset=True)
    for t in range(30):
        st = sc.get_state()
        wave.set_state(*st)
        sc.step_begin(None, *b0); sc.step_end()
        wave.step_fused(None, *b2)
        pts = sc.get_diag(points=True)[3]
        for x, z in zip(list(b0) + list(sc.get_state()), list(b2) + list(wave.get_state())):
            assert torch.equal(x, z), f"step {t}: an env with more than 16 candidate pairs differs from the wave-kernel scene"
    st = sc.exact_stats()
    # (the list instantiation has 16 candidate lanes too: it hands every one of these envs on to the wave-per-env kernel)
    # (... the first step through the list instantiation; every env above 16 points starts a heavy phase, whose launches send them there too)
    r = sc.exact_route()
    # (an overflow run: every step after the first is the rotated launch with three contacts per lane, which defers these envs again --
    #  they have no 48-point row -- to the list instantiation, which hands them on; MIR_EXACT_HEAVY set: the first step through the list
    #  instantiation, then a heavy phase whose launches send them to the wave-per-env kernel directly)
    assert r["wave_env_steps"] == 30 * n and r["list_env_steps"] + n * r["heavy_steps"] == 30 * n and r["big_steps"] + r["heavy_steps"] == 29, r
    assert st["overflow_env_steps"] == 30 * n and int(pts.min()) > 16
    assert torch.isfinite(sc.get_state()[0]).all() and float(sc.get_state()[0][:, 9].min()) > 0.05   # (the upper comb stays on the lower one)


def test_tight_loop_every_route_gives_the_same_trajectory(monkeypatch):
    """GenesisEnv.step back to back WITHOUT a host synchronisation between the steps (the scripted grasp at 4096 envs, 200 steps): the
    launches of consecutive steps overlap on the device -- the list launches and the first-half launches of an overflow run on the side
    stream, the next step's launch behind them through events.  Whatever route the overflow steps take -- list launches behind the bytes
    and the heavy phase (MIR_EXACT_BIG=0), the two-launch steps of an overflow run whenever the rows are there (2), or the library's
    own choice by whether the GPU was waiting for the call (1, the default) -- every observation of every step is the same bits.
    (Found a missing stream wait when the first-half launch moved to the side stream: the null stream is a stream.)"""
    from gym_genesis.env import GenesisEnv

    dev = torch.device("cuda", 0)
    pos, acts = _grasp_workload(B)
    quat = torch.tensor([[0.0, 0.0, 0.0, 1.0]]).repeat(B, 1)
    home = torch.tensor(HOME).repeat(B, 1)
    dacts = torch.as_tensor(acts, device=dev)
    ref = None
    for mode in ("0", "2", "1"):
        monkeypatch.setenv("MIR_EXACT_BIG", mode)
        env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False, exact_contacts=True)
        env.reset(seed=0)
        mir = env._env._mir
        mir.reset(pos, quat, home)
        mir.exact_stats(reset=True)
        out = []
        for t in range(dacts.shape[0]):
            o, r, term, _, _ = env.step(dacts[t])
            out.append(torch.cat([o["agent_pos"], o["environment_state"], r[:, None]], 1))
        torch.cuda.synchronize()
        out = torch.stack(out)
        st, route = mir.exact_stats(), mir.exact_route()
        print(f"\n[tight loop, MIR_EXACT_BIG={mode}] {st} {route}")
        assert st["overflow_env_steps"] > 1000
        if mode == "0":
            ref, ref_st = out, st
            assert route["big_steps"] == 0
        else:
            assert st == ref_st and (mode != "2" or route["big_steps"] > 0)
            bad = (out != ref).flatten(1).any(1)
            assert not bool(bad.any()), f"MIR_EXACT_BIG={mode}: first differing step {int(torch.nonzero(bad)[0])}"
        del env


def test_the_route_of_an_overflow_step_follows_what_the_caller_does_between_two_steps(monkeypatch):
    """The default (MIR_EXACT_BIG=1): a step of an overflow run is two launches -- second half, then the first half of the next step on the
    side stream -- when the caller spent at least 40 us between mir_step_end's return and mir_step_begin (the reference's expert: its policy
    and IK), and the list launches / heavy phase when it did not (those cost less GPU time).  The scripted grasp at 512 envs, once with
    nothing between the steps and once with 150 us of host work: the first takes no two-launch step, the second takes them through its
    overflow runs; both end in the same bits."""
    import time

    from gym_genesis.env import GenesisEnv

    monkeypatch.delenv("MIR_EXACT_BIG", raising=False)
    n = 512
    pos, acts = _grasp_workload(n)
    quat = torch.tensor([[0.0, 0.0, 0.0, 1.0]]).repeat(n, 1)
    home = torch.tensor(HOME).repeat(n, 1)
    finals, routes = [], []
    for pause in (0.0, 150e-6):
        env = GenesisEnv(task="cube_pick", robot="franka", num_envs=n, enable_pixels=False, exact_contacts=True)
        env.reset(seed=0)
        mir = env._env._mir
        mir.reset(pos, quat, home)
        mir.exact_stats(reset=True)
        dacts = torch.as_tensor(acts, device=mir.device)
        for t in range(dacts.shape[0]):
            env.step(dacts[t])
            if pause:
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < pause:
                    pass
        torch.cuda.synchronize()
        finals.append([x.clone() for x in mir.get_state()[:2]])
        routes.append((mir.exact_stats(), mir.exact_route()))
        del env
    (st0, r0), (st1, r1) = routes
    print(f"\n[route by the caller's gap] tight loop {r0}; 150 us between the steps {r1}")
    assert st0 == st1 and st0["overflow_env_steps"] > 100
    # (a busy host may stall the tight loop for 40 us once: not a single step is asked of it, only fewer than the loop that waits)
    assert r1["big_steps"] >= 2 and r0["big_steps"] < r1["big_steps"] and r0["big_steps"] <= 1, (r0, r1, st1)
    assert all(torch.equal(a, b) for a, b in zip(*finals))
