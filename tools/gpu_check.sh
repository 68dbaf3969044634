#!/bin/bash
# One GPU-box pass: the -m gpu tests, the driver's bench command, and the three terminated-hand-over modes side by side.
# Usage (from the repo root on the GPU box):  bash tools/gpu_check.sh [quick]
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "${1:-}" != "quick" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
  tail -40 gpurun_out/pytest_gpu.log
fi
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.out 2> gpurun_out/bench_driver.err
echo "bench rc=$?"
tail -c 6000 gpurun_out/bench_driver.out
tail -5 gpurun_out/bench_driver.err
for m in 0 1 2 3; do
  MIR_SYNC_MODE=$m timeout 300 python3 bench.py --gpus 1 --steps 2000 --warmup 50 --core-only > gpurun_out/bench_sync$m.out 2> gpurun_out/bench_sync$m.err
  echo "sync mode $m rc=$?"
  python3 - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_sync$m.out").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "hot_path_rate", "api_over_hot_path", "ms_per_step")}, d.get("roofline", {}).get("kernel_us"), d["config"]["terminated_sync_mode"])
except Exception as e:
    print("parse failed", e)
PY
done
