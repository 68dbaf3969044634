"""The scene-specialised instantiation of the 16-lane kernel (FEAT bit 2: the CubePick scene's sizes and options as literals,
csrc/mir_spec_pick.h) against the generic one (MIR_NO_SPEC=1 at mir_create): same source, same order of floating-point
operations, so every output of every launch -- fused, rotated, split in two, K-step rollout -- and the final state must agree
BIT FOR BIT, and a scene that does not match the literals must never get the specialised kernel."""
import numpy as np
import pytest
import torch

from gym_genesis.backend import models

pytestmark = pytest.mark.gpu
HOME = np.array(models.FRANKA_HOME, dtype=np.float32)


def _reset(sc, B, seed=0):
    rng = np.random.RandomState(seed)
    pos = np.stack([rng.uniform(0.45, 0.80, B), rng.uniform(-0.25, 0.25, B), np.full(B, 0.02)], 1).astype(np.float32)
    pos[::2, 2] = rng.uniform(0.12, 0.16, size=pos[::2].shape[0])  # every other cube falls onto the floor: contacts come and go
    pos[1::4, :2] = (0.55, 0.0)                                      # ... and some sit where the swinging hand finds them
    quat = np.tile(np.array([1, 0, 0, 0], np.float32), (B, 1))
    sc.reset(pos, quat, np.tile(HOME, (B, 1)))


def _pair(franka_spec, monkeypatch, B, split):
    from gym_genesis.backend.lib import MirScene

    monkeypatch.setenv("MIR_SPLIT_STEP", split)
    monkeypatch.delenv("MIR_NO_SPEC", raising=False)
    sc = MirScene(franka_spec, B)
    monkeypatch.setenv("MIR_NO_SPEC", "1")
    monkeypatch.setenv("MIR_SPLIT_STEP", "0")
    ref = MirScene(franka_spec, B)
    assert sc.spec_active and not ref.spec_active
    assert sc.split_step == int(split) and ref.split_step == 0
    return sc, ref


@pytest.mark.parametrize("B", [37, 4096])
@pytest.mark.parametrize("split", ["0", "1", "2"])
def test_specialised_launches_equal_generic_fused_steps_bit_for_bit(franka_spec, monkeypatch, split, B):
    sc, ref = _pair(franka_spec, monkeypatch, B, split)
    _reset(sc, B)
    _reset(ref, B)
    T = 120
    acts = torch.as_tensor(np.random.default_rng(5).uniform(-1, 1, (T, B, 9)).astype(np.float32), device=sc.device)
    b1 = (sc.empty(9), sc.empty(11), sc.empty(), sc.empty(dtype=torch.uint8))
    b2 = (ref.empty(9), ref.empty(11), ref.empty(), ref.empty(dtype=torch.uint8))
    contacts = 0
    for t in range(T):
        if t == 60:
            _reset(sc, B, seed=9); _reset(ref, B, seed=9)
        sc.step_begin(acts[t], *b1)
        host = sc.step_end()
        ref.step_fused(acts[t], *b2)
        assert np.array_equal(host, b2[3].cpu().numpy().astype(bool)), f"step {t}"
        for x, y in zip(b1, b2):
            assert torch.equal(x, y), f"step {t}"
        if t % 20 == 19:
            ref.set_diag(True); ref.step_fused(acts[t], *b2); contacts += int(ref.get_diag()[0].sum()); ref.set_diag(False)
            sc.step_fused(acts[t], *b1)   # (the same extra fused step on the specialised scene)
            for x, y in zip(b1, b2):
                assert torch.equal(x, y), f"extra step {t}"
    assert contacts > 0
    for x, y in zip(sc.get_state(), ref.get_state()):
        assert torch.equal(x, y)


def test_specialised_rollout_equals_generic_rollout_bit_for_bit(franka_spec, monkeypatch):
    B, K = 256, 40
    sc, ref = _pair(franka_spec, monkeypatch, B, "1")
    _reset(sc, B)
    _reset(ref, B)
    acts = torch.as_tensor(np.random.default_rng(6).uniform(-1, 1, (K, B, 9)).astype(np.float32), device=sc.device)
    r1 = torch.zeros((K, B, 22), device=sc.device)
    r2 = torch.zeros((K, B, 22), device=sc.device)
    sc.rollout(acts, r1)
    ref.rollout(acts, r2)
    assert torch.equal(r1, r2)
    for x, y in zip(sc.get_state(), ref.get_state()):
        assert torch.equal(x, y)


def test_scenes_that_differ_from_the_literals_run_the_generic_kernel(monkeypatch):
    from gym_genesis.backend.lib import MirScene

    monkeypatch.delenv("MIR_NO_SPEC", raising=False)
    assert MirScene(models.franka_cube_pick_scene().build(), 4).spec_active
    assert not MirScene(models.franka_cube_pick_scene(link_shape="box").build(), 4).spec_active      # no round geoms
    assert not MirScene(models.so101_cube_pick_scene().build(), 4).spec_active                         # another arm
    assert not MirScene(models.franka_cube_stack_scene().build(), 4).spec_active                       # the wave kernel's scene
    sb = models.franka_cube_pick_scene()
    spec = sb.build()
    spec.opt.iterations = 20                                                                             # same scene, another solver option
    assert not MirScene(spec, 4).spec_active
