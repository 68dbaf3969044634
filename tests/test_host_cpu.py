"""CPU-tier tests: C-ABI surface, golden fixtures derived from the reference's own logic, and
the host-side drop-in API (registry, GenesisEnv, task) on an oracle-backed test double.
No GPU compute is called here."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import torch

import gym_genesis
from gym_genesis import _gym
from gym_genesis.backend import lib as mirlib
from gym_genesis.backend import models, spec as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- C ABI
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mirigid.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(mir_\w+)\s*\(", hdr, flags=re.M))
    assert {"mir_create", "mir_destroy", "mir_reset", "mir_set_pd_targets", "mir_step", "mir_step_fused", "mir_step_packed",
            "mir_get_obs", "mir_get_state", "mir_set_state", "mir_get_links", "mir_last_error", "mir_version", "mir_render",
            "mir_visual_sizeof", "mir_autoreset"} <= declared
    assert {"mir_step_begin", "mir_step_end", "mir_debug_profile_step", "mir_debug_poison_lds"} <= declared
    lib = mirlib.load_library()
    for name in declared:
        assert hasattr(lib, name), f"libmirigid.so does not export {name}"
    # ... and nothing beyond the header: every exported mir_* symbol is declared (the launchers between the library's
    # translation units have hidden visibility)
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", mirlib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if ln.split()[-2:-1] == ["T"] and ln.split()[-1].startswith("mir_")}
    assert exported <= declared, f"exported but not declared in include/mirigid.h: {sorted(exported - declared)}"
    assert lib.mir_version() == S.MIR_VERSION
    assert lib.mir_spec_sizeof() == C.sizeof(S.MirSceneSpec)
    assert lib.mir_visual_sizeof() == C.sizeof(S.MirVisualSpec)


def test_builtin_step_calls_are_the_library_functions_bound_by_address():
    """gym_genesis.backend._mirfast (csrc/mir_pyfast.c) carries mir_step_prepare / mir_step_go / mir_step_end as CPython built-ins:
    it must be built, be bound to the loaded library by load_library(), and pass handles through untouched -- a null handle comes
    back as the library's own MIR_E_INVALID with its message, no GPU involved."""
    lib = mirlib.load_library()
    assert mirlib._fast is not None, "_mirfast.so is not built (make -C gym-genesis_amd/csrc)"
    f = mirlib._fast
    assert f.go(0, 0, 0) == -1 and b"null MirHandle" in lib.mir_last_error()
    assert f.end(None, 0) == -1 and f.prepare(0, 0, 0, 0, 0) == -1
    with pytest.raises(TypeError):
        f.go(0, 0)
    with pytest.raises(TypeError):
        f.go("handle", 0, 0)


def test_no_cpu_fallback_create_fails_loudly_without_gpu(franka_spec):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mirlib.MirError):
        mirlib.MirScene(franka_spec, 4)
    lib = mirlib.load_library()
    h = C.c_void_p()
    rc = lib.mir_create(C.byref(franka_spec), 4, 0, C.byref(h))
    assert rc == -4 and b"no usable HIP device" in lib.mir_last_error()  # MIR_E_NODEVICE


def test_spec_validation_errors(franka_spec):
    lib = mirlib.load_library()
    h = C.c_void_p()
    bad = S.MirSceneSpec.from_buffer_copy(franka_spec)
    bad.struct_size = 1
    assert lib.mir_create(C.byref(bad), 4, 0, C.byref(h)) == -1 and b"ABI" in lib.mir_last_error()
    bad = S.MirSceneSpec.from_buffer_copy(franka_spec)
    bad.opt.max_contacts = 1000
    assert lib.mir_create(C.byref(bad), 4, 0, C.byref(h)) == -2
    assert lib.mir_create(C.byref(franka_spec), 0, 0, C.byref(h)) == -1


# ---------------------------------------------------------------- golden fixtures (reference logic)
def test_reset_rng_stream_matches_golden_fixture(monkeypatch):
    import fake_scene
    from gym_genesis.tasks.franka import cube_pick

    monkeypatch.setattr(cube_pick, "MirScene", fake_scene.OracleScene)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "reset_rng.json")))
    for key, g in gold.items():
        if g["B"] > 64:
            continue
        task = cube_pick.FrankaCubePickBatch(False, 480, 640, g["B"], (1.0, 1.0), "global", True)
        task.seed(g["seed"])
        pos = task.sample_spawn()
        assert pos.dtype == np.float32
        assert np.array_equal(pos[:4], np.array(g["first_rows_f32"], np.float32)[:g["B"]])
        assert np.array_equal(pos[-1], np.array(g["last_row_f32"], np.float32))
        assert task._random.uniform(0.45, 0.80, size=(g["B"],))[0] == g["second_reset_x0_f64"]
    # SURVEY.md 8c-1 literal
    rs = np.random.RandomState(0)
    assert np.allclose(rs.uniform(0.45, 0.80, size=(4,)), [0.64208473, 0.70031628, 0.66096718, 0.64070911])


def test_spawn_draws_made_ahead_in_chunks_are_the_reference_stream(env):
    """tasks/spawn_ahead.py: the next reset's x / y blocks drawn a few values per env.step() (complete, partial, or not at all)
    are the values the reference's two uniform() calls draw (cube_pick.py:90-91); seed() drops what was drawn ahead."""
    from gym_genesis.tasks.spawn_ahead import UniformBlocksAhead

    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "reset_rng.json")))["seed0_B4"]
    for chunk, n_adv in ((1, 100), (3, 2), (5, 1), (1000, 1), (3, 0)):
        rs, ref = np.random.RandomState(11), np.random.RandomState(11)
        ah = UniformBlocksAhead([(0.45, 0.80, 7), (-0.25, 0.25, 7)])
        for rep in range(3):
            ah.start(rs)
            for _ in range(n_adv):
                if ah.left:
                    ah.advance(chunk)
            x, y = ah.take(rs)
            assert np.array_equal(x, ref.uniform(0.45, 0.80, size=(7,))) and np.array_equal(y, ref.uniform(-0.25, 0.25, size=(7,)))
        assert rs.uniform() == ref.uniform()      # both streams stand at the same place
    # through the task and the env.step closure: 3 envs -> 6 draws, one value per step would do; the closure draws >= 1024 at once
    task = env._env
    env.reset(seed=gold["seed"])
    ref = np.random.RandomState(gold["seed"])
    ref.uniform(size=6)                            # the reset above
    a = np.zeros((3, 9), np.float32)
    assert task._ahead.left == 6
    env.step(a)
    assert task._ahead.left == 0                  # drawn while the "kernel" ran
    want = np.stack([ref.uniform(0.45, 0.80, size=(3,)), ref.uniform(-0.25, 0.25, size=(3,)), np.full(3, 0.02)], 1).astype(np.float32)
    assert np.array_equal(task.sample_spawn(), want)
    env.step(a)                                    # the next block is drawn ahead again ...
    assert task._ahead.left == 0
    task.seed(gold["seed"])                        # ... and dropped by seed(): the stream starts over
    r2 = np.random.RandomState(gold["seed"])
    want = np.stack([r2.uniform(0.45, 0.80, size=(3,)), r2.uniform(-0.25, 0.25, size=(3,)), np.full(3, 0.02)], 1).astype(np.float32)
    assert np.array_equal(task.sample_spawn(), want)


def test_reference_constants():
    """Constants the reference spells out (file:line in models.py / cube_pick.py docstrings)."""
    assert models.FRANKA_HOME == (0.0, -0.4, 0.0, -2.2, 0.0, 2.0, 0.8, 0.04, 0.04)  # cube_pick.py:100
    assert models.FRANKA_JOINTS[7:] == ("finger_joint1", "finger_joint2")            # cube_pick.py:15-16
    sb = models.franka_cube_pick_scene()
    sp = sb.build()
    assert sp.opt.dt == 0.01                                                           # cube_pick.py:45
    cube = sp.body[sp.task.obj_body]
    assert tuple(cube.pos) == (0.65, 0.0, 0.02)                                        # cube_pick.py:53
    g = [sp.geom[i] for i in range(sp.ngeom) if sp.geom[i].body == sp.task.obj_body][0]
    assert tuple(g.size) == (0.02, 0.02, 0.02)                                         # 0.04^3 box, cube_pick.py:53
    assert sp.task.reward_z == 0.1 and sb.bodies[sp.task.eef_body]["name"] == "hand"   # cube_pick.py:134, :68
    assert [sp.task.grip_dof[i] for i in range(2)] == [7, 8]                           # cube_pick.py:142


# ---------------------------------------------------------------- drop-in API on the test double
@pytest.fixture
def env(monkeypatch):
    import fake_scene
    from gym_genesis.tasks.franka import cube_pick

    monkeypatch.setattr(cube_pick, "MirScene", fake_scene.OracleScene)
    from gym_genesis.env import GenesisEnv

    return GenesisEnv(task="cube_pick", robot="franka", num_envs=3, enable_pixels=False)


def test_env_reset_step_contract(env):
    B = 3
    obs, info = env.reset(seed=0)
    assert info == {"is_success": [False] * B}                                         # env.py:56
    assert set(obs) == {"agent_pos", "environment_state"}
    assert obs["agent_pos"].shape == (B, 9) and obs["environment_state"].shape == (B, 11)
    assert obs["agent_pos"].dtype == torch.float32
    assert env.observation_space["agent_pos"].shape == (9,) and env.action_space.shape == (9,)  # per-env spaces
    # reset consumed exactly one physics step (cube_pick.py:107): cube fell g dt^2 below the spawn height?  it rests on
    # the plane at z=0.02, so after one step it has sunk by less than a millimetre and reads (0,0,0,1) as set (cube_pick.py:94)
    es = obs["environment_state"].numpy()
    assert np.allclose(es[:, 3:7], [0, 0, 0, 1], atol=1e-6) and (np.abs(es[:, 2] - 0.02) < 1e-3).all()
    assert np.allclose(es[:, 0], [0.64208473, 0.70031628, 0.66096718][:B], atol=1e-6)  # golden x stream, seed 0
    # privileged features: diff = eef - cube, dist = |diff| (cube_pick.py:147-148)
    ap = obs["agent_pos"].numpy()
    assert np.allclose(es[:, 7:10], ap[:, :3] - es[:, :3], atol=1e-6)
    assert np.allclose(es[:, 10], np.linalg.norm(es[:, 7:10], axis=1), atol=1e-6)
    assert np.allclose(ap[:, 7:9], 0.04, atol=1e-4)
    # step with NumPy actions as the README loop does (README.md:34)
    act = np.stack([env.action_space.sample() for _ in range(B)])
    obs2, reward, terminated, truncated, info = env.step(act)
    assert isinstance(terminated, np.ndarray) and terminated.dtype == bool and terminated.shape == (B,)
    assert isinstance(truncated, np.ndarray) and truncated.dtype == bool and not truncated.any()  # env.py:65
    assert reward.shape == (B,) and reward.dtype == torch.float32
    assert np.array_equal(terminated, reward.numpy() == 1) and np.array_equal(info["is_success"].numpy(), terminated)
    assert obs2["agent_pos"] is not obs["agent_pos"]  # fresh tensors per step
    # ... and really fresh: the outputs of the NEXT call are allocated while this call's kernel runs (MirScene.step_fresh), so an
    # observation the caller keeps must not be overwritten by later steps, and info["is_success"] stays the step's own mask
    keep_obs, keep_rew, keep_succ = obs2["agent_pos"].clone(), reward.clone(), info["is_success"].clone()
    held_obs, held_rew, held_succ = obs2["agent_pos"], reward, info["is_success"]
    for _ in range(3):
        env.step(np.stack([env.action_space.sample() for _ in range(B)]))
    assert torch.equal(held_obs, keep_obs) and torch.equal(held_rew, keep_rew) and torch.equal(held_succ, keep_succ)
    assert info["is_success"].dtype == torch.bool
    # torch actions are accepted too (pick_cube_state.py:53)
    env.step(torch.as_tensor(act))
    with pytest.raises(ValueError):
        env.step(act[:2])


def test_env_accessors_and_errors(env):
    from gym_genesis.env import GenesisEnv

    assert env.render() is None                                                         # env.py:98 (pixels disabled)
    with pytest.raises(ValueError):
        env.get_cams()                                                                  # cube_pick.py:70-71
    env.reset(seed=1)
    robot, cube = env.get_robot(), env.get_cube()
    assert robot.get_dofs_position().shape == (3, 9) and cube.get_pos().shape == (3, 3)
    hand = robot.get_link("hand")
    assert torch.allclose(hand.get_pos(), env.get_obs()["agent_pos"][:, :3])
    z0 = float(cube.get_pos()[0, 2])
    env.push()                                                                          # env.py:59-60 -> scene.step()
    assert abs(float(cube.get_pos()[0, 2]) - z0) < 1e-3
    with pytest.raises(NotImplementedError) as ei:
        GenesisEnv(task="cube", robot="franka", num_envs=2)                             # env.py:122-123
    assert ei.value.args[0] == ("franka", "cube", True)
    assert GenesisEnv.metadata == {"render_modes": ["rgb_array"], "render_fps": 50}     # env.py:15


def test_pixels_contract_on_the_test_double(monkeypatch, tmp_path):
    """enable_pixels=True: obs keys, shapes, dtypes and error behaviour of the reference (cube_pick.py:70-84,159-180,
    env.py:97-98), host logic only (the images come from the oracle ray caster through the test double)."""
    import fake_scene
    from gym_genesis.env import GenesisEnv
    from gym_genesis.tasks.franka import cube_pick

    monkeypatch.setattr(cube_pick, "MirScene", fake_scene.OracleScene)
    B, H, W = 2, 24, 32
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=True, observation_height=H, observation_width=W,
                     camera_capture_mode="per_env", record_video=True)
    obs, _ = env.reset(seed=0)
    assert set(obs) == {"agent_pos", "pixels"}                                           # strip_environment_state default
    assert tuple(obs["pixels"].shape) == (B, H, W, 3) and obs["pixels"].dtype == torch.uint8
    assert env.observation_space["pixels"].shape == (H, W, 3) and env.observation_space["pixels"].dtype == np.uint8
    frame = env.render()                                                                 # env.py:98: cam.render()[0]
    assert isinstance(frame, np.ndarray) and frame.shape == (H, W, 3) and frame.dtype == np.uint8
    assert env.get_cams() is env._env.cam                                                # cube_pick.py:69-72
    env2 = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=True, observation_height=H, observation_width=W,
                      camera_capture_mode="global", strip_environment_state=False, record_video=True)
    obs2, _ = env2.reset(seed=0)
    assert set(obs2) == {"agent_pos", "environment_state", "pixels"} and tuple(obs2["pixels"].shape) == (H, W, 3)
    with pytest.raises(ValueError):
        GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=True, camera_capture_mode="bogus")  # cube_pick.py:177-178
    # the camera sits at (3.5, 0, 2.5) looking at (0, 0, 0.5) with fov 30 (cube_pick.py:57-62)
    cam = env._env.cam
    assert cam.pos == (3.5, 0.0, 2.5) and cam.lookat == (0.0, 0.0, 0.5) and cam.fov == 30.0 and cam.res == (W, H)
    # save_video (env.py:70-79): with record_video=True the recording started at reset() (cube_pick.py:109-110) holds the global renders since -- the one
    # env.render() above, two more here -- and is written as Motion-JPEG in an mp4 (tasks/video.py); the env goes on afterwards
    from gym_genesis.tasks.video import read_mjpeg_mp4
    env.step(np.zeros((B, 9), np.float32))
    f2 = env.render()
    f3 = env.render()
    path = str(tmp_path / "episode.mp4")
    with pytest.warns(UserWarning, match="stops the camera recording"):
        env.save_video(save_video=True, file_name=path, fps=30)
    frames, fps, wh = read_mjpeg_mp4(path)
    assert len(frames) == 3 and fps == 30.0 and wh == (W, H) and all(f.shape == (H, W, 3) for f in frames)
    for got, want in zip(frames, (frame, f2, f3)):   # (JPEG at quality 90: a few grey levels on these flat-shaded images)
        assert np.abs(got.astype(int) - want.astype(int)).mean() < 4.0
    env.save_video(save_video=False, file_name=str(tmp_path / "never.mp4"))          # (save_video=False: nothing happens, env.py:72)
    assert not (tmp_path / "never.mp4").exists()
    with pytest.warns(UserWarning, match="no recording is running"):                  # (the recording was stopped above: a warning, the loop goes on)
        env._env.cam.stop_recording("again.mp4")
    assert not os.path.exists("again.mp4")
    env.step(np.zeros((B, 9), np.float32))
    # the `global` pixels observation IS a global render: every step of env2 added a frame; a .gif works too
    env2.step(np.zeros((B, 9), np.float32))
    with pytest.warns(UserWarning):
        env2.save_video(save_video=True, file_name=str(tmp_path / "episode.gif"), fps=20)
    from PIL import Image
    assert Image.open(str(tmp_path / "episode.gif")).n_frames == 2
    # without record_video (the default: recorded frames stay alive, which costs the README loop 6-25 %) save_video says what to do
    env3 = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=True, observation_height=H, observation_width=W)
    env3.reset(seed=0)
    with pytest.warns(UserWarning, match="record_video=True"):
        env3.save_video(save_video=True, file_name=str(tmp_path / "none.mp4"))
    assert not (tmp_path / "none.mp4").exists()


def test_seeded_reset_is_deterministic(env):
    a, _ = env.reset(seed=42)
    b, _ = env.reset(seed=42)
    assert torch.equal(a["environment_state"], b["environment_state"])
    c, _ = env.reset()  # continues the stream
    assert not torch.equal(a["environment_state"][:, :2], c["environment_state"][:, :2])


def test_registry_ids_and_defaults(monkeypatch):
    import fake_scene
    from gym_genesis.tasks.franka import cube_pick

    monkeypatch.setattr(cube_pick, "MirScene", fake_scene.OracleScene)
    if _gym.HAVE_GYMNASIUM:
        pytest.skip("real gymnasium registry")
    reg = _gym._REGISTRY
    assert set(reg) >= {"gym_genesis/CubePick-v0", "gym_genesis/CubeStack-v0"}          # __init__.py:4,22
    e = reg["gym_genesis/CubePick-v0"]
    assert e["max_episode_steps"] == 200 and e["entry_point"] == "gym_genesis.env:GenesisEnv"
    assert e["kwargs"] == dict(task="cube_pick", robot="so101", enable_pixels=False, num_envs=10, observation_height=480,
                               observation_width=640, env_spacing=(1.0, 1.0), camera_capture_mode="global",
                               strip_environment_state=True)                            # __init__.py:8-18
    env = gym_genesis.make("gym_genesis/CubePick-v0", robot="franka", num_envs=2)
    env.reset(seed=0)
    act = np.zeros((2, 9), np.float32)
    for t in range(200):
        obs, rew, term, trunc, info = env.step(act)
    assert trunc is True  # TimeLimit(200) replaces truncated at the 200th step (gymnasium behaviour)
    assert env.unwrapped.num_envs == 2


# ---------------------------------------------------------------- SO-101 pick (registry default robot)
def test_so101_pick_contract(monkeypatch):
    import fake_scene
    from gym_genesis.tasks.so101 import cube_pick as so

    monkeypatch.setattr(so, "MirScene", fake_scene.OracleScene)
    env = gym_genesis.make("gym_genesis/CubePick-v0", num_envs=2).unwrapped if not _gym.HAVE_GYMNASIUM else None
    if env is None:
        pytest.skip("real gymnasium registry")
    assert type(env._env).__name__ == "CubePick" and env.robot == "so101"               # registry default robot (__init__.py:10)
    obs, info = env.reset(seed=0)
    rs = np.random.RandomState(0)
    x = rs.uniform(-0.32, -0.28, size=(2,)); y = rs.uniform(-0.05, 0.05, size=(2,))     # so101/cube_pick.py:61-62
    es = obs["environment_state"].numpy()
    assert obs["agent_pos"].shape == (2, 8) and es.shape == (2, 11)                     # eef 7 + gripper 1 ; 11
    assert np.allclose(es[:, 0], x, atol=1e-6) and np.allclose(es[:, 1], y, atol=1e-6)
    assert np.allclose(es[:, 2], models.ISLAND_TOP_Z + 0.021, atol=1e-6)                # no physics step in reset (:81)
    assert np.allclose(es[:, 3:7], [1, 0, 0, 0])                                        # quat (1,0,0,0) (:68)
    assert env.observation_space["agent_pos"].shape == (6,) and env.action_space.shape == (6,)  # declared spaces (:45-56)
    obs, reward, terminated, truncated, info = env.step(np.zeros((2, 6), np.float32))
    assert terminated.all() and (reward == 1).all()  # cube z ~0.72 > 0.1: the reference's threshold fires at once (:112)
    assert env.get_robot() is env._env.so_101 and env.get_cube() is env._env.cube


def test_so101_stack_spawn_sampling_equals_the_scalar_rejection_loop():
    """tasks/so101/cube_stack_batch.py draws the rejection-sampled cube pairs as arrays; positions and the state of the stream
    afterwards must be those of the reference's scalar loop (cube_stack_batch.py:72-103)."""
    import types

    import numpy as np

    from gym_genesis.tasks.so101.cube_stack_batch import CubeStackBatch as SO101CubeStackBatch

    def scalar(Bg, r, z):
        p1, p2 = np.zeros((Bg, 3)), np.zeros((Bg, 3))
        for e in range(Bg):
            while True:
                x1 = r.uniform(-0.3, -0.1)
                y1 = r.uniform(-0.1, 0.1)
                x2 = r.uniform(-0.3, -0.1)
                y2 = r.uniform(-0.1, 0.1)
                if ((x2 - x1) ** 2 + (y2 - y1) ** 2) ** 0.5 >= 0.06:
                    p1[e], p2[e] = (x1, y1, z), (x2, y2, z)
                    break
        cols = [p1, p2]
        for _ in range(3):
            x = r.uniform(-0.35, 0.0, size=(Bg,))
            y = r.uniform(-0.2, 0.2, size=(Bg,))
            cols.append(np.stack([x, y, np.full(Bg, z)], axis=1))
        return np.stack(cols, axis=1).astype(np.float32)

    for seed, Bg in ((0, 1), (1, 7), (2, 300), (3, 2048)):
        ra, rb = np.random.RandomState(seed), np.random.RandomState(seed)
        stub = types.SimpleNamespace(global_num_envs=Bg, _random=ra, island_top_z=0.7)
        for _ in range(3):  # consecutive resets continue one stream
            got = SO101CubeStackBatch.sample_spawn(stub)
            want = scalar(Bg, rb, 0.7 + 0.02 + 0.001)
            assert got.shape == (Bg, 5, 3) and np.array_equal(got, want)
        assert ra.random_sample() == rb.random_sample()


# ---------------------------------------------------------------- the getters as the reference's experts call them (round 6)
def test_getters_take_envs_idx_like_genesis(env):
    """`robot.get_qpos(envs_idx=np.arange(B))`, `robot.get_link("hand").get_pos(envs_idx=torch.arange(B))`
    (/root/reference/examples/franka/stack_cube_state.py:57,76; examples/so_101/collect_task_stack_cube_batch.py:73,86,90): NumPy, torch and
    list indices, the identity without a gather, a subset in the order asked for."""
    B = 3
    env.reset(seed=0)
    robot, link = env.get_robot(), env.get_robot().get_link("hand")
    q, p, quat = robot.get_qpos(), link.get_pos(), link.get_quat()
    assert q.shape == (B, 9) and p.shape == (B, 3) and quat.shape == (B, 4)
    for idx in (np.arange(B), torch.arange(B), list(range(B))):
        assert torch.equal(robot.get_qpos(envs_idx=idx), q) and torch.equal(link.get_pos(envs_idx=idx), p) and torch.equal(link.get_quat(envs_idx=idx), quat)
    sub = [2, 0]
    assert torch.equal(robot.get_qpos(envs_idx=sub), q[sub]) and torch.equal(link.get_pos(envs_idx=np.array(sub)), p[sub])
    assert torch.equal(robot.get_dofs_position(dofs_idx_local=[7, 8], envs_idx=torch.tensor(sub)), q[sub][:, 7:9])
    assert torch.equal(robot.get_dofs_velocity(envs_idx=sub), robot.get_dofs_velocity()[sub])
    assert torch.equal(env.get_cube().get_pos(envs_idx=sub), env.get_cube().get_pos()[sub])


def test_inverse_kinematics_never_writes_a_callers_tensor(env):
    """rows addressed through `envs_idx` are scattered into tensors of the wrapper's own; a full-batch argument handed in beside
    them -- the caller's tensor -- is read, never edited (an all-zero quaternion row used to be turned into the identity IN PLACE)."""
    B = 3
    obs, _ = env.reset(seed=0)
    robot = env.get_robot()
    eef = robot.get_link("hand")
    pos_sub = obs["environment_state"][[2, 0], :3] + torch.tensor([0.0, 0.0, 0.2])
    quat_full = torch.tensor([[0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0]])   # row 1 is not addressed
    keep = quat_full.clone()
    v0 = quat_full._version
    q = robot.inverse_kinematics(link=eef, pos=pos_sub, quat=quat_full, envs_idx=[2, 0])
    assert q.shape == (2, 9) and torch.isfinite(q).all()
    assert torch.equal(quat_full, keep) and quat_full._version == v0
    # scattered quaternions: the rows nobody addressed become the identity inside the wrapper's own tensor
    q2 = robot.inverse_kinematics(link=eef, pos=pos_sub, quat=keep[[2, 0]], envs_idx=[2, 0])
    assert torch.allclose(q, q2)


def test_inverse_kinematics_one_launch_equals_the_scatter_gather_wrapper(env):
    """robot.inverse_kinematics hands its arguments to mir_inverse_kinematics_rows as they come -- by row of `envs_idx`, by env, one
    quaternion for all, seeds for the arm's columns (gym_genesis/tasks/views.py) -- instead of scattering them to batch rows around the
    full-batch launch; on the test double both routes run the oracle's solver: the same rows come back, for every way of addressing."""
    from gym_genesis.tasks import views

    obs, _ = env.reset(seed=0)
    robot = env.get_robot()
    eef = robot.get_link("hand")
    B = obs["agent_pos"].shape[0]
    pos_full = obs["environment_state"][:, :3] + torch.tensor([0.0, 0.0, 0.2])
    one = torch.tensor([0.0, 1.0, 0.0, 0.0])
    quat_full = one.repeat(B, 1) + torch.tensor([[0.0, 0.0, 0.05 * e, 0.0] for e in range(B)])
    seed_full = robot.get_qpos() + 0.02 * torch.arange(B * 9, dtype=torch.float32).reshape(B, 9) / (B * 9)
    sub = [2, 0]   # (fewer rows than envs: an argument with B rows is addressed by env, as in Genesis)
    cases = [dict(pos=pos_full, quat=one.expand(B, -1), envs_idx=torch.arange(B)),
             dict(pos=pos_full, quat=quat_full, init_qpos=seed_full),
             dict(pos=pos_full[sub], quat=quat_full[sub], envs_idx=sub),
             dict(pos=pos_full, quat=one, envs_idx=np.array(sub), init_qpos=seed_full[sub]),
             dict(pos=pos_full[sub], quat=quat_full, envs_idx=torch.tensor(sub), init_qpos=seed_full),
             dict(pos=pos_full.numpy(), quat=None, envs_idx=[1])]
    calls = []
    real = env._env._mir.inverse_kinematics_rows
    env._env._mir.inverse_kinematics_rows = lambda *a, **k: (calls.append(a[5]), real(*a, **k))[1]
    try:
        for i, kw in enumerate(cases):
            views.IK_ROWS = True
            q1, e1 = robot.inverse_kinematics(link=eef, return_error=True, **kw)
            views.IK_ROWS = False
            kw2 = dict(kw)
            if kw2.get("quat") is not None and torch.as_tensor(kw2["quat"]).numel() == 4:
                n = B if kw2.get("envs_idx") is None else len(kw2["envs_idx"])
                kw2["quat"] = one.reshape(1, 4).repeat(n, 1)
            q0, e0 = robot.inverse_kinematics(link=eef, return_error=True, **kw2)
            assert q1.shape == q0.shape and torch.equal(q1, q0) and torch.equal(e1, e0), f"case {i}"
    finally:
        views.IK_ROWS = True
        del env._env._mir.inverse_kinematics_rows
    # the flags each case was launched with: pos by env | one quaternion; pos + quat + init by env; all by row; ...
    assert calls == [1 | 4, 1 | 2 | 8, 0, 1 | 4, 2 | 8, 1], calls


def _example(rel):
    import importlib.util

    spec = importlib.util.spec_from_file_location(os.path.basename(rel)[:-3], os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("robot,rel,steps,dim", [("franka", "examples/franka/stack_cube_state.py", 391, 9),
                                                 ("so101", "examples/so_101/collect_task_stack_cube_batch.py", 360, 6)])
def test_reference_stack_experts_run_on_the_test_double(monkeypatch, robot, rel, steps, dim):
    """The reference's two batched stack experts, restated constant for constant (examples/), drive the env end to end on the CPU test
    double: path lengths as in the reference (3 x 77 + 2 x 80 joint targets for the Franka, 5 x 70 + 10 for the SO-101), finite
    states, the gripper schedule of the grasp stage (open, then closing over its last five targets)."""
    import fake_scene
    from gym_genesis.env import GenesisEnv
    from gym_genesis.tasks import stack_common

    monkeypatch.setattr(stack_common, "MirScene", fake_scene.OracleScene)
    ex = _example(rel)
    env = GenesisEnv(task="cube_stack", robot=robot, num_envs=2, enable_pixels=False, strip_environment_state=False)
    obs, _ = env.reset(seed=1)
    policy = ex.expert_policy if robot == "franka" else ex.expert_policy_v2
    grasp = policy(env.get_robot(), obs, "grasp")
    assert len(grasp) == (77 if robot == "franka" else 70) and grasp[0].shape == (2, dim)
    opened, closed = (0.04, -0.02) if robot == "franka" else (0.5, 0.1)
    assert all(abs(float(a[0, -1]) - opened) < 1e-6 for a in grasp[:-5])
    assert abs(float(grasp[-5][0, -1]) - opened) < 1e-6 and opened > float(grasp[-1][0, -1]) > closed   # alpha = 0 .. 4 / 5
    out = ex.run_episode(env, obs)
    acts = out[2] if robot == "franka" else out[1]
    assert acts.shape == (steps, 2, dim) and all(np.isfinite(x).all() for x in out)
    with pytest.raises(ValueError):
        policy(env.get_robot(), obs, "no such stage")
