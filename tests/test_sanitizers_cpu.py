"""CPU sanitizer tier (no GPU sanitizers exist on this pool): the float64 oracle and the product's host-compiled code --- the two
scene compilers, the spec emitter, the _mirfast module, the convex narrowphase header built for the host --- run under
AddressSanitizer + UndefinedBehaviorSanitizer in a child process (tests/asan_worker.py) with libasan preloaded.  Targets:
`make -C oracle asan`, `make -C gym-genesis_amd/csrc asan-host`.  Skipped where gcc has no libasan."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _lib(name):
    try:
        p = subprocess.check_output(["gcc", f"-print-file-name={name}"], text=True).strip()
    except Exception:  # noqa: BLE001
        return None
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_and_host_code_run_clean_under_asan_and_ubsan():
    asan = _lib("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan.so here")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "gym-genesis_amd", "csrc"), "asan-host"], stdout=subprocess.DEVNULL)
    env = dict(os.environ)
    env.update({"LD_PRELOAD": asan, "ORC_SANITIZE": "1", "OMP_NUM_THREADS": "2",
                # (CPython itself leaks at exit and uses its own allocator tricks: leak checking off; everything else fatal)
                "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=23",
                "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1:exitcode=24"})
    r = subprocess.run([sys.executable, os.path.join(HERE, "asan_worker.py")], env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, f"sanitizer worker failed (rc {r.returncode}):\n{tail}"
    assert "SANITIZER_WORKER_OK" in r.stdout, tail
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, tail
    print(r.stdout.strip().splitlines()[-1])
