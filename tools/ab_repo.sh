#!/bin/bash
# Same-box A/B of this tree against another checkout of the repo (built, e.g. tools/probes/ab/base_repo = `git archive <rev>` + make):
# alternates `bench.py --steps 2000 --core-only` of the two trees and prints value / us per step / hot-path rate / kernel us.
# Usage (GPU box, repo root): bash tools/ab_repo.sh <other_repo_dir> [rounds]
set -u
O=$1; R=${2:-3}
run() {
  (cd $2 && python3 bench.py --gpus 1 --steps 2000 --warmup 50 --core-only 2>/dev/null | tail -1) | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
r = d.get('roofline', {})
print(sys.argv[1], round(d['value'] / 1e6, 2), 'M |', round(d['ms_per_step'] * 1e3, 3), 'us/step | median region', round(d.get('value_median_region', 0) / 1e6, 2), 'M | hot', round(d.get('hot_path_rate', 0) / 1e6, 2), 'M | kernel', round(r.get('kernel_us', 0), 3), 'us | fused', round(r.get('fused_kernel_us', 0) or 0, 3), '| early', d.get('early_mask'))" $1
}
for r in $(seq $R); do
  run base $O
  run new .
done
