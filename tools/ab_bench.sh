#!/bin/bash
# Same-box A/B of two builds of libmirigid.so (run on the GPU box from the repo root): alternates the two libraries three times
# under `bench.py --steps 2000 --core-only` and prints value / us per step / hot-path rate / kernel us for each run.
# Usage: bash tools/ab_bench.sh <a.so> <b.so>   (box-to-box differences are larger than most single changes; this is not)
set -u
A=$1; B=$2
L=gym-genesis_amd/csrc/libmirigid.so
cp $L /tmp/ab_keep.so
run() {
  python3 bench.py --gpus 1 --steps 2000 --warmup 50 --core-only 2>&1 | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print(sys.argv[1], round(d['value'] / 1e6, 2), 'M |', round(d['ms_per_step'] * 1e3, 3), 'us/step | hot', round(d['hot_path_rate'] / 1e6, 2), 'M | kernel', round(d['roofline']['kernel_us'], 3), 'us')" $1
}
for r in 1 2 3; do
  cp $A $L; run A
  cp $B $L; run B
done
cp /tmp/ab_keep.so $L
