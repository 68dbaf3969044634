"""Probe: what would a launch that stays resident over several steps cost per step at best?

Every rotated launch of the headline loop waits for the one env of 4096 that runs a third Newton iteration (first workgroup out at
11.9 us, median 14.4, last 18.7: profiles/r4/pick_phase_profile.txt), then pays the launch boundary.  A resident launch would let
every workgroup go through its steps at its own pace.  mir_debug_resident_steps is that on the real step body with raw-launch
semantics (actions of all steps resident, nothing handed to the host): per-step time over 1000 steps against 1000 back-to-back
rotated launches and 1000 back-to-back fused launches on the same box, HIP events on the launching stream.

Decision rule of the verdict: <= 17 us per step: write the opt-in design; > 19 us: close the lead.  A watchdog thread ends the process
with a non-zero code if a launch does not come back (nothing is re-executed, nothing restarted)."""
import ctypes as C
import os
import sys
import threading

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "gym-genesis_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gym_genesis.env import GenesisEnv  # noqa: E402


def main() -> int:
    threading.Timer(240.0, lambda: (sys.stderr.write("resident_steps: watchdog\n"), os._exit(3))).start()
    B, N = 4096, 1000
    dev = torch.device("cuda", 0)
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
    mir = env._env._mir
    mir.lib.mir_debug_resident_steps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    mir.lib.mir_debug_resident_steps.restype = C.c_int
    acts = torch.empty((N, B, 9), dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=torch.Generator(device=dev).manual_seed(1234))
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    res = {}

    def timed(name, fn, reps=3):
        ts = []
        for _ in range(reps):
            env.reset(seed=0)
            torch.cuda.synchronize()
            e0, e1 = ev(), ev()
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / N)
        res[name] = sorted(ts)[len(ts) // 2]
        print(f"{name:34s} {res[name]:7.2f} us per step (regions {np.round(ts, 2).tolist()})", flush=True)

    def resident():
        mir._check(mir.lib.mir_debug_resident_steps(mir.h, C.c_void_p(acts.data_ptr()), N, mir._stream()))

    def resident_chunks(k):
        def f():
            for i in range(0, N, k):
                mir._check(mir.lib.mir_debug_resident_steps(mir.h, C.c_void_p(acts[i].data_ptr()), min(k, N - i), mir._stream()))
        return f

    def fused():
        for i in range(N):
            env._env.step_raw(acts[i])

    timed("rotated launches (today's path)", lambda: mir.rotated_launches(acts, N))
    timed("fused launches", fused)
    timed("resident, 1000 steps in 1 launch", resident)
    timed("resident, 16 steps per launch", resident_chunks(16))
    timed("resident, 4 steps per launch", resident_chunks(4))
    ratio = res["resident, 1000 steps in 1 launch"] / res["rotated launches (today's path)"]
    print(f"resident / rotated = {ratio:.3f}")
    os._exit(0)


if __name__ == "__main__":
    sys.exit(main())
