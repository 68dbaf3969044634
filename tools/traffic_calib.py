"""GPU probe: 20 launches of mir_debug_copy_rows over 64 MiB (the step kernel's access shape: 4 B per lane, 64-thread workgroups).
Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (tools/collect_profiles.sh): each launch moves exactly 64 MiB in
and 64 MiB out, which calibrates the two counters for this access width."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]
from gym_genesis.backend import lib  # noqa: E402

L = lib.load_library()
L.mir_debug_copy_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
L.mir_debug_copy_rows.restype = C.c_int
n = 16 * 1024 * 1024  # floats = 64 MiB
src = torch.rand(n, device="cuda")
dst = torch.empty_like(src)
for _ in range(20):
    assert L.mir_debug_copy_rows(src.data_ptr(), dst.data_ptr(), n, 0, torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
assert torch.equal(src, dst)
print(f"copied {n * 4} bytes x 20")
