"""Developer tool, and the command profiled by tools/profile_ik.sh: bench.py's `ik` leg alone (robot.inverse_kinematics from the home pose
to 0.25 m above the cube, 4096 envs, 200 calls) followed by the reference expert's five stages (one IK call per env.step)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gym-genesis_amd"))
import torch
import bench
dev = torch.device("cuda", 0)
out = bench.ik_bench(torch, dev, 200)
print(json.dumps({k: v for k, v in out.items() if k != "workload"}))
