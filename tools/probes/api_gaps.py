"""Run under `rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/probes/api_gaps.py run` to record an env.step
loop, then `python3 tools/probes/api_gaps.py parse <dir>` for the step kernel's durations and the idle gaps between consecutive
launches (start of launch n+1 minus end of launch n) -- i.e. whether the loop is bound by the GPU or by the host coming back late."""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd")]
    import torch
    from gym_genesis.env import GenesisEnv
    B = 4096
    env = GenesisEnv(task="cube_pick", robot="franka", num_envs=B, enable_pixels=False)
    env.reset(seed=0)
    dev = env._env.device
    g = torch.Generator(device=dev).manual_seed(0)
    acts = [torch.empty((B, 9), device=dev).uniform_(-1, 1, generator=g) for _ in range(25)]
    for t in range(1500):
        obs, reward, terminated, truncated, info = env.step(acts[t % 25])
        if terminated.any() or truncated.any():
            pass
    torch.cuda.synchronize()
else:
    import csv
    import numpy as np
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "mir_step_kernel<" in r["Kernel_Name"]]
    s = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64)
    e = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64)
    o = np.argsort(s); s, e = s[o], e[o]
    s, e = s[200:], e[200:]
    dur = (e - s) / 1e3
    gap = (s[1:] - e[:-1]) / 1e3
    per = (s[1:] - s[:-1]) / 1e3
    q = lambda x: " ".join(f"{np.percentile(x, p):6.2f}" for p in (5, 25, 50, 75, 95))
    print(f"{len(s)} rotated launches; percentiles 5 25 50 75 95 (us)")
    print(f"  duration          mean {dur.mean():6.2f} | {q(dur)}")
    print(f"  gap to the next   mean {gap.mean():6.2f} | {q(gap)}")
    print(f"  start to start    mean {per.mean():6.2f} | {q(per)}")
