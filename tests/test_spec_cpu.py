"""The scene-specialised instantiation of the 16-lane kernel reads its sizes and options from csrc/mir_spec_pick.h, which the
library's own scene compiler emits for the CubePick scene (tools/gen_scene_spec.py).  The committed header must be what the
compiler emits NOW: after a change to models.py or mir_compile.cpp it would otherwise silently stop matching (mir_create then runs
the generic instantiation: correct, but slower).  Host only."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_scene_spec", os.path.join(ROOT, "tools", "gen_scene_spec.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_scene_constants_are_what_the_compiler_emits_for_the_headline_scene():
    mod = _gen()
    text = mod.generate()
    assert open(mod.HEADER).read() == text, "stale mir_spec_pick.h: run tools/gen_scene_spec.py and rebuild"
    for field in ("nbody = 13", "nv = 15", "gj_split = 9", "has_convex = 1", "fk_free_leaf = 1"):
        assert f"static constexpr int {field};" in text


def test_emit_spec_refuses_scenes_of_the_wave_kernel_and_small_buffers():
    import ctypes as C

    from gym_genesis.backend import models
    from gym_genesis.backend.lib import load_library

    lib = load_library()
    buf = C.create_string_buffer(1 << 14)
    stack = models.franka_cube_stack_scene().build()
    assert lib.mir_debug_emit_spec(C.byref(stack), b"X", buf, len(buf)) == -2  # MIR_E_CAPACITY: 39 dofs
    pick = models.franka_cube_pick_scene().build()
    assert lib.mir_debug_emit_spec(C.byref(pick), b"X", buf, 16) == -2
    n = lib.mir_debug_emit_spec(C.byref(pick), b"X", buf, len(buf))
    assert n > 0 and buf.value.decode().startswith("struct X {")
    # another scene of the 16-lane kernel emits other constants: the box-link Panda has no round geoms
    box = models.franka_cube_pick_scene(link_shape="box").build()
    assert lib.mir_debug_emit_spec(C.byref(box), b"X", buf, len(buf)) > 0 and "has_convex = 0;" in buf.value.decode()
