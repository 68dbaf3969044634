#!/bin/bash
# Run on the GPU box from the repo root: kernel-trace stats and issue-side counters (separate --pmc passes) of
# bench.py's `ik` leg (tools/ik_time.py).  Summaries -> gpurun_out/ik_kernel_stats.csv, gpurun_out/ik_sq_counters.json, gpurun_out/ik_time.log.
set -u
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ik -- python3 $R/tools/ik_time.py > $R/gpurun_out/ik_under_rocprof.log 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_INSTS_BRANCH SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_ik$i -- python3 $R/tools/ik_time.py > $R/gpurun_out/pmc_ik$i.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for f in glob.glob("gpurun_out/pmc_ik[0-9]*/*/*counter_collection.csv"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "mir_ik_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        v = sorted(v.values())
        out[k] = {"launches": len(v), "median_per_launch": v[len(v) // 2]}
json.dump(out, open("gpurun_out/ik_sq_counters.json", "w"), indent=1, sort_keys=True)
PY
find gpurun_out/prof_ik -name "*kernel_stats.csv" -exec cp {} gpurun_out/ik_kernel_stats.csv \;
rm -rf gpurun_out/pmc_ik*/ gpurun_out/prof_ik
python3 tools/ik_time.py > gpurun_out/ik_time.log 2>&1
head -6 gpurun_out/ik_kernel_stats.csv; cat gpurun_out/ik_sq_counters.json | head -30; cat gpurun_out/ik_time.log | tail -1
