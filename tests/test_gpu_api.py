"""GPU tests of the env.step() host hand-over (mir_step_begin / mir_step_end): in every sync mode the NumPy `terminated` that
GenesisEnv.step returns equals the device mask of the same launch bit for bit, the physics is untouched (bit-identical to
mir_step_fused), and each call hands out a fresh array (gym_genesis/env.py:61-69 of the reference)."""
import numpy as np
import pytest
import torch

from gym_genesis.backend import models

pytestmark = pytest.mark.gpu
HOME = np.array(models.FRANKA_HOME, dtype=np.float32)


def _reset(sc, B, seed=0):
    rng = np.random.RandomState(seed)
    pos = np.stack([rng.uniform(0.45, 0.80, B), rng.uniform(-0.25, 0.25, B), np.full(B, 0.02)], 1).astype(np.float32)
    pos[::2, 2] = rng.uniform(0.12, 0.16, size=pos[::2].shape[0])  # every other cube starts above the reward threshold and falls
    quat = np.tile(np.array([0, 0, 0, 1], np.float32), (B, 1))
    sc.reset(pos, quat, np.tile(HOME, (B, 1)))


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("B", [1, 5, 4096])
def test_step_begin_end_equals_step_fused(franka_spec, monkeypatch, mode, B):
    from gym_genesis.backend.lib import MirScene

    monkeypatch.setenv("MIR_SYNC_MODE", str(mode))
    sc = MirScene(franka_spec, B)
    monkeypatch.delenv("MIR_SYNC_MODE")
    ref = MirScene(franka_spec, B)
    assert sc.sync_mode == mode and ref.sync_mode == 3
    _reset(sc, B)
    _reset(ref, B)
    acts = torch.as_tensor(np.random.default_rng(3).uniform(-1, 1, (25, B, 9)).astype(np.float32), device=sc.device)
    b1 = (sc.empty(9), sc.empty(11), sc.empty(), sc.empty(dtype=torch.uint8))
    b2 = (ref.empty(9), ref.empty(11), ref.empty(), ref.empty(dtype=torch.uint8))
    seen_true = seen_false = False
    for t in range(25):
        sc.step_begin(acts[t], *b1)
        host = sc.step_end()
        ref.step_fused(acts[t], *b2)
        assert host.dtype == np.bool_ and host.shape == (B,)
        assert np.array_equal(host, b1[3].cpu().numpy().astype(bool)), f"step {t}: host mask differs from the device mask"
        for x, y in zip(b1, b2):
            assert torch.equal(x, y)
        seen_true |= bool(host.any())
        seen_false |= bool((~host).any())
    assert seen_false and (seen_true or B == 1)
    for x, y in zip(sc.get_state(), ref.get_state()):
        assert torch.equal(x, y)


def test_step_end_without_begin_is_an_error_and_an_open_step_is_closed_by_the_next_begin_or_reset(franka_spec):
    """A step left open (the caller raised between begin and end) must not brick the scene: the next mir_step_begin / mir_reset
    waits for its bytes and drops them.  The physics is that of two ordinary steps."""
    from gym_genesis.backend.lib import MirError, MirScene

    sc, ref = MirScene(franka_spec, 8), MirScene(franka_spec, 8)
    _reset(sc, 8)
    _reset(ref, 8)
    with pytest.raises(MirError):
        sc.step_end()
    bufs = (sc.empty(9), sc.empty(11), sc.empty(), sc.empty(dtype=torch.uint8))
    rb = (ref.empty(9), ref.empty(11), ref.empty(), ref.empty(dtype=torch.uint8))
    sc.step_begin(None, *bufs)
    sc.step_begin(None, *bufs)      # closes the open one first
    host = sc.step_end()
    ref.step_fused(None, *rb)
    ref.step_fused(None, *rb)
    assert host.shape == (8,) and np.array_equal(host, bufs[3].cpu().numpy().astype(bool))
    for x, y in zip(sc.get_state(), ref.get_state()):
        assert torch.equal(x, y)
    sc.step_begin(None, *bufs)
    _reset(sc, 8)                   # ... and so does a reset
    with pytest.raises(MirError):
        sc.step_end()
    sc.step_begin(None, *bufs)
    assert sc.step_end().shape == (8,)


def test_terminated_tag_survives_every_other_user_of_the_sequence_counter(franka_spec):
    """The tag of the host-visible terminated bytes has its own counter: 2001 null round trips between two steps (bench.py does
    exactly that) used to leave the next launch with the tag already in the buffer, and mir_step_end returned the PREVIOUS step's
    mask before the kernel had written anything.  Step 1 leaves all-False bytes, step 2 must deliver all-True."""
    from gym_genesis.backend.lib import MirScene

    B = 4096
    sc = MirScene(franka_spec, B)
    bufs = (sc.empty(9), sc.empty(11), sc.empty(), sc.empty(dtype=torch.uint8))
    for n_null in (2001, 3, 126, 127, 1):
        _reset(sc, B, seed=1)
        q = sc.get_state()[0]
        q[:, 11] = 0.02                                   # every cube on the floor: terminated all False
        sc.set_state(qpos=q)
        sc.step_begin(None, *bufs)
        assert not sc.step_end().any()
        sc.null_roundtrip_us(n_null)
        q[:, 11] = 0.5                                    # every cube released well above the threshold: all True
        sc.set_state(qpos=q)
        sc.step_begin(None, *bufs)
        host = sc.step_end()
        assert host.all(), f"{n_null} null round trips: stale terminated bytes were accepted ({int(host.sum())} of {B} True)"
    for _ in range(300):                                  # more steps than there are tags
        sc.step_begin(None, *bufs)
        host = sc.step_end()
        assert np.array_equal(host, bufs[3].cpu().numpy().astype(bool))


def test_env_step_returns_fresh_host_masks_every_call():
    from gym_genesis.env import GenesisEnv

    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=64, enable_pixels=False)
    env.reset(seed=0)
    a = torch.zeros((64, 9), device=env._env.device)
    kept = []
    for t in range(5):
        obs, reward, terminated, truncated, info = env.step(a)
        assert isinstance(terminated, np.ndarray) and terminated.dtype == np.bool_ and truncated.dtype == np.bool_
        assert np.array_equal(terminated, (reward == 1).cpu().numpy()) and torch.equal(info["is_success"].cpu(), torch.as_tensor(terminated))
        kept.append((terminated, terminated.copy(), obs["agent_pos"], obs["agent_pos"].clone()))
    for arr, snap, ten, tsnap in kept:  # earlier results are not overwritten by later steps
        assert np.array_equal(arr, snap) and torch.equal(ten, tsnap)
    assert len({id(k[0]) for k in kept}) == 5


@pytest.mark.parametrize("which", ["pick", "stack"])
@pytest.mark.parametrize("iterations", [0, 1])
def test_two_wave_kernels_with_a_solver_that_never_or_once_iterates(which, iterations):
    """The single-step instantiations run two waves per workgroup that meet at four barriers; the last one sits inside the first
    Newton iteration, or behind the loop when no iteration needs the Hessian.  A solver capped at 0 or 1 iterations (and envs
    without any constraint) must take the second route without hanging, and the K-step rollout (one wave, no barriers) must
    still agree bit for bit."""
    from gym_genesis.backend.lib import MirScene

    sb = models.franka_cube_pick_scene() if which == "pick" else models.franka_cube_stack_scene()
    sb.opt["iterations"] = iterations
    spec = sb.build()
    B = 64
    sc, sr = MirScene(spec, B), MirScene(spec, B)
    nfree = 1 if which == "pick" else 5
    rng = np.random.RandomState(3)
    pos = np.zeros((B, nfree, 3), np.float32)
    pos[..., 0] = rng.uniform(0.45, 0.8, (B, nfree)) if which == "pick" else rng.uniform(-0.3, 0.3, (B, nfree))
    pos[..., 1] = rng.uniform(-0.25, 0.25, (B, nfree))
    pos[..., 2] = 0.5 if which == "pick" else 1.5  # in the air: the first steps have no contact at all
    quat = np.tile(np.array([1, 0, 0, 0], np.float32), (B, nfree, 1))
    home = np.tile(np.array(models.FRANKA_HOME, np.float32), (B, 1))
    for s in (sc, sr):
        s.reset(pos.reshape(B, -1) if nfree == 1 else pos, quat.reshape(B, -1) if nfree == 1 else quat, home)
    K = 12
    g = torch.Generator(device=sc.device).manual_seed(0)
    acts = torch.from_numpy(home).to(sc.device) + torch.empty((K, B, 9), device=sc.device).uniform_(-0.5, 0.5, generator=g)
    stride = sc.agent_dim + sc.env_dim + 2
    rows1 = torch.zeros((K, B, stride), device=sc.device)
    for k in range(K):
        sc.step_packed(acts[k], rows1[k])
    rowsK = torch.zeros((K, B, stride), device=sc.device)
    sr.rollout(acts, rowsK)
    torch.cuda.synchronize()
    assert torch.isfinite(rows1).all()
    if which == "pick":  # (the wave kernel's plain rollout IS K single-step launches)
        assert torch.equal(rows1, rowsK)
    assert all(torch.equal(a, b) for a, b in zip(sc.get_state(), sr.get_state()))
    assert int(sc.get_diag()[2].max()) <= iterations


@pytest.mark.parametrize("split", ["0", "1", "2"])
def test_split_step_survives_everything_that_touches_the_state_between_two_steps(franka_spec, monkeypatch, split):
    """GenesisEnv.step's launch leaves the action-independent half of the NEXT step behind (MIR_SPLIT_STEP: 1 = one rotated launch,
    2 = two launches, 0 = off).  That half is only valid for the state the launch left: a reset (whole batch or masked), a state
    write, a plain step, a K-step rollout, a render or a kinematics query in between must either keep it valid or make the next
    step fall back to the fused launch.  300 steps with all of those mixed in, against a twin scene that only ever runs fused
    launches: every output and the final state bit-identical."""
    from gym_genesis.backend.lib import MirScene
    from gym_genesis.backend.spec import make_camera

    B = 64
    monkeypatch.setenv("MIR_SPLIT_STEP", split)
    sc = MirScene(franka_spec, B)
    monkeypatch.setenv("MIR_SPLIT_STEP", "0")
    ref = MirScene(franka_spec, B)
    assert sc.split_step == int(split) and ref.split_step == 0
    _reset(sc, B)
    _reset(ref, B)
    g = np.random.default_rng(11)
    acts = torch.as_tensor(g.uniform(-1, 1, (300, B, 9)).astype(np.float32), device=sc.device)
    b1 = (sc.empty(9), sc.empty(11), sc.empty(), sc.empty(dtype=torch.uint8))
    b2 = (ref.empty(9), ref.empty(11), ref.empty(), ref.empty(dtype=torch.uint8))
    rows = torch.zeros((4, B, 22), device=sc.device)
    cam = make_camera(32, 24, (3.5, 0, 2.5), (0, 0, 0.5), 30)
    vis = models.franka_cube_pick_scene().visual()
    for t in range(300):
        k = t % 37
        if k == 5:      # reset of the whole batch
            _reset(sc, B, seed=t); _reset(ref, B, seed=t)
        elif k == 11:   # masked reset
            mask = torch.as_tensor(g.integers(0, 2, B).astype(np.uint8), device=sc.device)
            pos = np.stack([g.uniform(.45, .8, B), g.uniform(-.25, .25, B), np.full(B, .02)], 1).astype(np.float32)
            quat = np.tile(np.array([1, 0, 0, 0], np.float32), (B, 1)); home = np.tile(np.array(models.FRANKA_HOME, np.float32), (B, 1))
            for s in (sc, ref):
                s.reset(pos, quat, home, env_mask=mask)
        elif k == 17:   # state written back (qvel perturbed)
            q, v, tg, ws = sc.get_state()
            for s in (sc, ref):
                s.set_state(qpos=q, qvel=v * 0.5, target=tg, warmstart=ws)
        elif k == 21:   # a plain fused step and a rollout in between
            for s, b in ((sc, b1), (ref, b2)):
                s.step_fused(acts[t], *b)
                s.rollout(acts[t:t + 4].contiguous() if t + 4 <= 300 else acts[:4].contiguous(), rows)
        elif k == 27:   # read-only visitors: image, link poses, observation
            sc.render(cam, vis); sc.get_links(); sc.get_obs()
        elif k == 31:   # PD targets set from outside, then a step without an action
            for s in (sc, ref):
                s.set_pd_targets(acts[t] * 0.3)
            sc.step_begin(None, *b1); h = sc.step_end()
            ref.step_fused(None, *b2)
            assert np.array_equal(h, b2[3].cpu().numpy().astype(bool))
        sc.step_begin(acts[t], *b1)
        host = sc.step_end()
        ref.step_fused(acts[t], *b2)
        assert np.array_equal(host, b2[3].cpu().numpy().astype(bool)), f"step {t}"
        for x, y in zip(b1, b2):
            assert torch.equal(x, y), f"step {t}"
    for x, y in zip(sc.get_state(), ref.get_state()):
        assert torch.equal(x, y)


def test_env_step_over_builtins_equals_env_step_over_ctypes_and_a_closed_scene_is_an_error(monkeypatch):
    """GenesisEnv.step's flat closure exists twice (tasks/fast_step.py): over the _mirfast built-ins and over ctypes (taken when the
    module is missing).  Same launches, same results bit for bit; after MirScene.close() both raise instead of touching freed memory."""
    from gym_genesis.backend import lib as mirlib
    from gym_genesis.env import GenesisEnv

    assert mirlib._fast is not None, "_mirfast.so did not travel / is not built"
    e1 = GenesisEnv(task="cube_pick", robot="franka", num_envs=64, enable_pixels=False)
    monkeypatch.setattr(mirlib, "_fast", None)
    e2 = GenesisEnv(task="cube_pick", robot="franka", num_envs=64, enable_pixels=False)
    monkeypatch.undo()
    assert e1.step.__name__ == "fast_step_builtin" and e2.step.__name__ == "fast_step"
    e1.reset(seed=3); e2.reset(seed=3)
    g = torch.Generator(device=e1._env.device).manual_seed(0)
    for t in range(40):
        a = torch.empty((64, 9), device=e1._env.device).uniform_(-1, 1, generator=g)
        r1, r2 = e1.step(a), e2.step(a)
        assert all(torch.equal(r1[0][k], r2[0][k]) for k in r1[0]) and torch.equal(r1[1], r2[1])
        assert np.array_equal(r1[2], r2[2]) and np.array_equal(r1[3], r2[3]) and torch.equal(r1[4]["is_success"], r2[4]["is_success"])
    for e in (e1, e2):
        e._env._mir.close()
        with pytest.raises(mirlib.MirError):
            e.step(a)


def test_back_to_back_rotated_launches_with_outputs_equal_fused_steps(franka_spec, monkeypatch):
    """bench.py times the rotated kernel through mir_debug_rotated_launches with the device outputs of a GenesisEnv.step launch: those
    launches must BE steps -- same state, same four outputs as fused launches fed the same actions -- and must not disturb the
    output registration (mir_step_prepare) a step closure made ahead of its next call."""
    from gym_genesis.backend.lib import MirScene

    B, K, n = 64, 5, 12
    monkeypatch.setenv("MIR_SPLIT_STEP", "1")  # (rotated launches exist in this mode only, whatever the environment asks for)
    sc, ref = MirScene(franka_spec, B), MirScene(franka_spec, B)
    assert sc.split_step == 1
    _reset(sc, B); _reset(ref, B)
    acts = torch.as_tensor(np.random.default_rng(5).uniform(-1, 1, (K, B, 9)).astype(np.float32), device=sc.device)
    b1 = (sc.empty(9), sc.empty(11), sc.empty(), sc.empty(dtype=torch.uint8))
    b2 = (ref.empty(9), ref.empty(11), ref.empty(), ref.empty(dtype=torch.uint8))
    ahead = (sc.empty(9), sc.empty(11), sc.empty(), sc.empty(dtype=torch.uint8))
    sc.step_prepare_ptrs([t.data_ptr() for t in ahead])
    sc.rotated_launches(acts, n, outputs=b1)   # (no first half is waiting after a reset: one fused step on acts[0] goes first)
    ref.step_fused(acts[0], *b2)
    for i in range(n):
        ref.step_fused(acts[i % K], *b2)
    for x, y in zip(b1, b2):
        assert torch.equal(x, y)
    for x, y in zip(sc.get_state(), ref.get_state()):
        assert torch.equal(x, y)
    sc.rotated_launches(acts, 3)              # without outputs; the first half left by the launches above is used
    for i in range(3):
        ref.step_fused(acts[i % K], *b2)
    sc.step_go_ptr(acts[3].data_ptr())         # the registration made before the debug launches is still there
    host = np.empty(B, np.bool_)
    sc.step_end_ptr(host.ctypes.data)
    ref.step_fused(acts[3], *b2)
    for x, y in zip(ahead, b2):
        assert torch.equal(x, y)
    assert np.array_equal(host, b2[3].cpu().numpy().astype(bool))


def test_task_step_begin_between_two_env_steps_does_not_leave_the_closure_with_unwritten_outputs():
    """The library holds ONE registration of output pointers (mir_step_prepare) and two Python callers make them: the env.step
    closure and task.step_begin.  They share the scene's slot cache, so whichever registered last is what the next launch uses
    and returns."""
    from gym_genesis.env import GenesisEnv

    B = 32
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
    env.reset(seed=0)
    task = env._env
    a = torch.as_tensor(np.random.default_rng(0).uniform(-1, 1, (6, B, 9)).astype(np.float32), device=task.device)
    env.step(a[0])
    task.step_begin(a[1])
    task.step_end()
    obs, reward, terminated, _, _ = env.step(a[2])
    agent, envst, rew, term = task._mir.get_obs()
    assert torch.equal(obs["agent_pos"], agent) and torch.equal(obs["environment_state"], envst) and torch.equal(reward, rew)
    assert np.array_equal(terminated, term.cpu().numpy().astype(bool))
    task.step(a[3])
    obs, reward, terminated, _, _ = env.step(a[4])
    agent, envst, rew, term = task._mir.get_obs()
    assert torch.equal(obs["agent_pos"], agent) and torch.equal(obs["environment_state"], envst)


def test_exception_inside_env_step_leaves_the_env_usable(monkeypatch):
    """An exception between launch and wait (out of memory in the allocation of the next outputs, KeyboardInterrupt) closes the
    step before it propagates; the next env.step works and the physics has advanced by exactly the launched step."""
    from gym_genesis.env import GenesisEnv

    B = 16
    env, ref = (GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False) for _ in range(2))
    env.reset(seed=0)
    ref.reset(seed=0)
    a = torch.zeros((B, 9), device=env._env.device)
    env.step(a)
    ref.step(a)
    real = np.zeros
    calls = {"n": 0}

    def boom(*args, **kw):
        calls["n"] += 1
        raise MemoryError("injected")

    monkeypatch.setattr("gym_genesis.tasks.fast_step.np.zeros", boom, raising=True)
    # (the closure bound np.zeros at creation: rebuild it so that the injected failure is the one it calls)
    env.step = env._env.make_fast_step()
    with pytest.raises(MemoryError):
        env.step(a)
    monkeypatch.setattr("gym_genesis.tasks.fast_step.np.zeros", real, raising=True)
    env.step = env._env.make_fast_step()
    ref.step(a)
    obs, _, terminated, _, _ = env.step(a)
    robs, _, rterm, _, _ = ref.step(a)
    assert calls["n"] == 1
    assert torch.equal(obs["agent_pos"], robs["agent_pos"]) and np.array_equal(terminated, rterm)
    env.reset()


def test_pinned_host_inputs_are_copied_unless_they_come_from_staged(franka_spec):
    """A caller's pinned tensor may be overwritten as soon as the call returns (the kernel must not read it in place); only
    MirScene.staged() buffers, whose reuse an event guards, are read over PCIe by the reset kernel."""
    from gym_genesis.backend.lib import MirScene

    B = 2048
    sc = MirScene(franka_spec, B)
    _reset(sc, B)
    q0 = sc.get_state()[0].cpu()
    pinned = q0.clone().pin_memory()
    for _ in range(20):
        sc.set_state(qpos=pinned)
        pinned.fill_(float("nan"))      # the caller reuses its buffer at once
        assert torch.equal(sc.get_state()[0].cpu(), q0)
        pinned.copy_(q0)
    st = sc.staged(np.zeros((B, 3), np.float32))
    assert st.is_pinned() and st.data_ptr() in sc._staged_ptrs


def test_launch_kinds_mixed_with_everything_that_touches_the_state_stay_bit_identical(franka_spec, monkeypatch):
    """What a launch carries over to the next one -- the scratch row of a rotated launch, the wave kernel's cached poses -- is only
    good for the state that launch stored: resets (whole batch or masked), state writes, rollouts, rotated / split launches and
    renders in between must either refresh it or mark it stale.  300 steps with all of those mixed in on a scene that uses rotated
    launches where it can, against a twin that runs one fused launch per step (MIR_SPLIT_STEP=0): every output and the final state
    bit-identical.  (Round 3's pose cache of the 16-lane kernel, which this test was written for, is gone.)"""
    from gym_genesis.backend.lib import MirScene
    from gym_genesis.backend.spec import make_camera

    B = 64
    monkeypatch.setenv("MIR_SPLIT_STEP", "1")
    sc = MirScene(franka_spec, B)
    monkeypatch.setenv("MIR_SPLIT_STEP", "0")
    ref = MirScene(franka_spec, B)
    _reset(sc, B)
    _reset(ref, B)
    g = np.random.default_rng(12)
    acts = torch.as_tensor(g.uniform(-1, 1, (300, B, 9)).astype(np.float32), device=sc.device)
    b1 = (sc.empty(9), sc.empty(11), sc.empty(), sc.empty(dtype=torch.uint8))
    b2 = (ref.empty(9), ref.empty(11), ref.empty(), ref.empty(dtype=torch.uint8))
    rows1 = torch.zeros((4, B, 22), device=sc.device)
    rows2 = torch.zeros((4, B, 22), device=sc.device)
    cam = make_camera(32, 24, (3.5, 0, 2.5), (0, 0, 0.5), 30)
    vis = models.franka_cube_pick_scene().visual()
    for t in range(300):
        k = t % 41
        if k == 5:      # reset of the whole batch
            _reset(sc, B, seed=t); _reset(ref, B, seed=t)
        elif k == 9:    # masked reset: the other envs keep their cached poses
            mask = torch.as_tensor(g.integers(0, 2, B).astype(np.uint8), device=sc.device)
            pos = np.stack([g.uniform(.45, .8, B), g.uniform(-.25, .25, B), np.full(B, .02)], 1).astype(np.float32)
            quat = np.tile(np.array([1, 0, 0, 0], np.float32), (B, 1)); home = np.tile(HOME, (B, 1))
            for s in (sc, ref):
                s.reset(pos, quat, home, env_mask=mask)
        elif k == 14:   # state written back with another qpos: the cached poses are those of the old one
            q, v, tg, ws = sc.get_state()
            q2 = q.clone(); q2[:, :7] += 0.01
            for s in (sc, ref):
                s.set_state(qpos=q2, qvel=v, target=tg, warmstart=ws)
        elif k == 19:   # a rollout launch in between
            sc.rollout(acts[:4].contiguous(), rows1); ref.rollout(acts[:4].contiguous(), rows2)
            assert torch.equal(rows1, rows2)
        elif k in (23, 24, 25):   # rotated launches (GenesisEnv.step's) between fused ones
            sc.step_begin(acts[t], *b1); h = sc.step_end()
            ref.step_fused(acts[t], *b2)
            assert np.array_equal(h, b2[3].cpu().numpy().astype(bool))
        elif k == 30:   # read-only visitors (the render refreshes the pose buffer through another launch)
            sc.render(cam, vis); sc.get_links(); sc.get_obs()
        elif k == 35:   # plain physics steps
            sc.step(3); ref.step(3)
        sc.step_fused(acts[t], *b1)
        ref.step_fused(acts[t], *b2)
        for x, y in zip(b1, b2):
            assert torch.equal(x, y), f"step {t}"
    for x, y in zip(sc.get_state(), ref.get_state()):
        assert torch.equal(x, y)


def test_early_terminated_bytes_equal_the_integrated_mask(franka_spec, monkeypatch):
    """The terminated bytes of a mir_step_begin launch leave from inside the solver loop once a convexity bound says the object's height
    cannot reach the threshold any more (csrc/mir_model.h: term_bound_ok).  Against a twin scene that always waits for the integrator
    (MIR_NO_EARLY_MASK=1): the host masks agree step for step -- random actions, cubes
    falling through the threshold, cubes thrown up through it, hands pushing cubes -- the kernel's own check counts no workgroup
    whose early bytes differed from the integrated state, and most workgroups did send early."""
    from gym_genesis.backend.lib import MirScene

    B = 1024
    monkeypatch.setenv("MIR_SPLIT_STEP", "1")
    sc = MirScene(franka_spec, B)
    monkeypatch.setenv("MIR_NO_EARLY_MASK", "1")
    ref = MirScene(franka_spec, B)
    sc.set_diag(True)
    g = np.random.default_rng(21)
    acts = torch.as_tensor(g.uniform(-1, 1, (64, B, 9)).astype(np.float32), device=sc.device)
    b1 = (sc.empty(9), sc.empty(11), sc.empty(), sc.empty(dtype=torch.uint8))
    b2 = (ref.empty(9), ref.empty(11), ref.empty(), ref.empty(dtype=torch.uint8))
    seen_true = seen_false = 0
    sc.early_mask_stats(reset=True)
    for phase in range(4):
        _reset(sc, B, seed=phase); _reset(ref, B, seed=phase)          # (every other cube starts above the threshold and falls through it)
        if phase >= 2:                                                  # cubes thrown upwards from the floor: they cross it on the way up
            q, v, tg, ws = sc.get_state()
            v2 = v.clone(); v2[1::4, 11] = 2.5 + 0.5 * phase
            for s in (sc, ref):
                s.set_state(qpos=q, qvel=v2, target=tg, warmstart=ws)
        for t in range(60):
            a = acts[(17 * phase + t) % 64]
            sc.step_begin(a, *b1); h1 = sc.step_end()
            ref.step_begin(a, *b2); h2 = ref.step_end()
            assert np.array_equal(h1, h2), f"phase {phase} step {t}"
            assert np.array_equal(h1, b1[3].cpu().numpy().astype(bool))
            for x, y in zip(b1, b2):
                assert torch.equal(x, y)
            seen_true += int(h1.sum()); seen_false += int((~h1).sum())
    early, bad = sc.early_mask_stats()
    assert bad == 0
    assert seen_true > 100 and seen_false > 100
    assert early > 0.3 * 4 * 60 * (B // 4)   # (workgroup-launches that sent early; the rest converge at the first gradient and store after the integrator)
    for x, y in zip(sc.get_state(), ref.get_state()):
        assert torch.equal(x, y)


@pytest.mark.parametrize("task,robot", [("cube_pick", "franka"), ("cube_pick", "so101"), ("cube_stack", "franka"), ("cube_stack", "so101")])
def test_numpy_actions_are_read_in_place_and_equal_device_actions(task, robot):
    """GenesisEnv.step(numpy array) -- how the reference is driven (env.py:61, action_space.sample()) -- stages the action in pinned
    memory that the launch reads in place (MirScene.stage_action).  Same trajectory, bit for bit, as the same actions given as device
    tensors; the caller may scribble over its array as soon as step() returns; lists and CPU tensors take the same route."""
    from gym_genesis.env import GenesisEnv

    B = 256
    a = GenesisEnv(task=task, robot=robot, num_envs=B)
    b = GenesisEnv(task=task, robot=robot, num_envs=B)
    a.reset(seed=3)
    b.reset(seed=3)
    dim = a.action_space.shape[-1]
    rng = np.random.default_rng(9)
    for t in range(40):
        act = rng.uniform(-1, 1, (B, dim)).astype(np.float32)
        if robot == "so101" and t % 2:
            act = act.astype(np.float64)  # (any dtype NumPy converts)
        dev_act = torch.as_tensor(act.astype(np.float32), device=a._env.device)
        host_in = act.copy() if t % 3 else (torch.from_numpy(act.astype(np.float32)) if t % 2 else act.tolist())
        oa, ra, ta, _, _ = a.step(host_in)
        if isinstance(host_in, np.ndarray):
            host_in[:] = 123.0  # the step has its own copy
        ob, rb, tb, _, _ = b.step(dev_act)
        assert np.array_equal(ta, tb)
        assert torch.equal(ra, rb) and torch.equal(oa["agent_pos"], ob["agent_pos"]) and torch.equal(oa["environment_state"], ob["environment_state"]), f"step {t}"
    with pytest.raises((ValueError, RuntimeError)):  # (the SO-101 task reshapes first: torch's error)
        a.step(np.zeros((B, dim + 1), np.float32))
    a.step(np.zeros((B, dim), np.float32))  # (the env stays usable)
    #  tensors that are not plain float32 tensors on the device: a Parameter on the device, a float64 device tensor, a CPU
    # tensor that requires grad -- all accepted, like the reference's permissive action handling
    b.step(np.zeros((B, dim), np.float32))  # (in step with `a` again)
    act = torch.zeros((B, dim), dtype=torch.float32, device=a._env.device)
    for odd in (torch.nn.Parameter(act.clone()), act.double(), torch.zeros((B, dim), requires_grad=True)):
        oa, ra, ta, _, _ = a.step(odd)
        ob, rb, tb, _, _ = b.step(act)
        assert np.array_equal(ta, tb) and torch.equal(oa["agent_pos"], ob["agent_pos"])


def test_row_all_reduce_gives_every_lane_the_same_bits():
    """The 16-lane kernel's solver takes its decisions -- converged? line search finished? step accepted? -- in every lane of an env
    from sums over the env's 16 lanes: the lanes must hold the SAME sum, to the bit (mir_dev.h: gsum; with the rotation butterfly it
    replaced, two lanes of a row could differ in the last bit, and an env's lanes stopped their line search at different evaluations).
    4096 rows of wide dynamic range and mixed signs: identical bits across each row, and the float64 sum within float32 rounding."""
    import ctypes as C

    from gym_genesis.backend import lib

    L = lib.load_library()
    L.mir_debug_row_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_void_p]
    L.mir_debug_row_sum.restype = C.c_int
    rng = np.random.default_rng(0)
    n = 4096
    x = (rng.normal(size=(n, 16)) * 10.0 ** rng.uniform(-6, 6, (n, 16))).astype(np.float32)
    x[::7, 5:] *= -1.0
    dev = torch.device("cuda", torch.cuda.current_device())
    xin, out = torch.as_tensor(x, device=dev), torch.empty((n, 16), dtype=torch.float32, device=dev)
    assert L.mir_debug_row_sum(C.c_void_p(xin.data_ptr()), C.c_void_p(out.data_ptr()), n, dev.index, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    o = out.cpu().numpy()
    bits = o.view(np.uint32)
    assert (bits == bits[:, :1]).all(), f"{int((bits != bits[:, :1]).any(1).sum())} rows hold different sums in different lanes"
    ref = x.astype(np.float64).sum(1)
    assert np.abs(o[:, 0] - ref).max() <= 16 * 6e-8 * np.abs(x.astype(np.float64)).sum(1).max()
    assert (np.abs(o[:, 0] - ref) <= 16 * 6e-8 * np.abs(x.astype(np.float64)).sum(1)).all()
