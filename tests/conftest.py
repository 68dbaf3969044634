import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gym-genesis_amd")
for p in (PKG, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def franka_spec():
    from gym_genesis.backend import models

    return models.franka_cube_pick_scene().build()


@pytest.fixture(autouse=True)
def _poison_lds(request):
    """GPU tests start from LDS full of NaN patterns (mir_debug_poison_lds): a step kernel that reads a slot before writing
    it would otherwise usually find the plausible leftovers of the previous launch there."""
    if "gpu" in request.keywords and _has_gpu():
        import ctypes as C

        import torch
        from gym_genesis.backend import lib

        L = lib.load_library()
        L.mir_debug_poison_lds.argtypes = [C.c_int, C.c_void_p]
        L.mir_debug_poison_lds.restype = C.c_int
        assert L.mir_debug_poison_lds(torch.cuda.current_device(), C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
        torch.cuda.synchronize()
    yield
