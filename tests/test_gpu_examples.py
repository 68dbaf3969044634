"""The caller's side of the hot path: the expert data-collection scripts under examples/ (shaped like the reference's
examples/franka/pick_cube_state.py and stack_cube_state.py) run end to end on the device -- batched IK -> env.step -> reward --
and write the LeRobot-named features."""
import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", "franka", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("script,env_dim,min_frac", [("pick_cube_state", 11, 0.8), ("stack_cube_state", 14, 0.7)])
def test_expert_collection(tmp_path, monkeypatch, script, env_dim, min_frac):
    out = str(tmp_path / "d.npz")
    # (the pick script's default stages are the reference's own constants, measured in tests/test_ref_expert.py; the schedule that
    #  keeps the fingers off the floor is the one whose episodes are counted here)
    argv = [script, "--num-envs", "32", "--out", out] + (["--episodes", "1", "--stages", "tuned"] if script.startswith("pick") else [])
    monkeypatch.setattr(sys, "argv", argv)
    kept = _load(script).main()
    assert kept >= int(min_frac * 32), kept
    d = np.load(out)
    n = d["action"].shape[0]
    assert n == kept * (200 if script.startswith("pick") else 391)   # (the reference's stack expert: 3 x 77 + 2 x 80 joint targets)
    assert d["observation.state"].shape == (n, 9) and d["observation.environment_state"].shape == (n, env_dim)
    assert d["episode_index"].max() == kept - 1 and np.isfinite(d["observation.state"]).all()


def test_image_expert_collection_writes_one_video_per_successful_env(tmp_path, monkeypatch):
    """examples/franka/pick_cube_image.py (the reference's pick_cube_image.py: the same expert with per-env pixels): the frames of the
    whole batch stay on the device, the envs that lifted the cube become one Motion-JPEG .mp4 each + the LeRobot-named features."""
    sys.path.insert(0, os.path.join(ROOT, "gym-genesis_amd"))
    from gym_genesis.tasks.video import read_mjpeg_mp4

    out = str(tmp_path / "img")
    monkeypatch.setattr(sys, "argv", ["pick_cube_image", "--num-envs", "8", "--episodes", "1", "--stages", "tuned", "--height", "96", "--width", "128",
                                      "--out", out])
    kept = _load("pick_cube_image").main()
    assert kept >= 6, kept
    d = np.load(os.path.join(out, "episodes.npz"))
    assert d["action"].shape == (kept * 200, 9) and d["observation.state"].shape == (kept * 200, 9) and d["episode_index"].max() == kept - 1
    vids = sorted(os.listdir(os.path.join(out, "videos")))
    assert vids == [f"episode_{k:06d}.mp4" for k in range(kept)]
    frames, fps, wh = read_mjpeg_mp4(os.path.join(out, "videos", vids[0]))
    assert len(frames) == 200 and fps == 60.0 and wh == (128, 96)
    # the arm moves and the cube goes up: first and last frame differ, consecutive ones hardly
    f = np.stack(frames).astype(int)
    assert np.abs(f[-1] - f[0]).mean() > 5 * np.abs(f[101] - f[100]).mean() > 0


def test_step_outputs_are_fresh_tensors_on_the_device():
    """GenesisEnv.step() hands out new tensors every call (the next call's outputs are allocated while the kernel runs):
    what the caller keeps from one step is not touched by later steps."""
    import torch

    sys.path.insert(0, os.path.join(ROOT, "gym-genesis_amd"))
    from gym_genesis.env import GenesisEnv

    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=16, enable_pixels=False)
    env.reset(seed=0)
    g = torch.Generator(device="cuda").manual_seed(0)
    acts = torch.empty((6, 16, 9), device="cuda").uniform_(-1, 1, generator=g)
    obs, rew, term, trunc, info = env.step(acts[0])
    held = (obs["agent_pos"], obs["environment_state"], rew, info["is_success"])
    kept = [t.clone() for t in held]
    for k in range(1, 6):
        o2, *_ = env.step(acts[k])
        assert o2["agent_pos"].data_ptr() != held[0].data_ptr()
    assert all(torch.equal(a, b) for a, b in zip(held, kept))
    assert info["is_success"].dtype == torch.bool and term.dtype == bool
