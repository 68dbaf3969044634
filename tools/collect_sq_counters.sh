#!/bin/bash
# Run on the GPU box from the repo root: issue-side counters of mir_step_kernel (separate --pmc passes).
set -u
R=$(pwd)
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_sq$i -- python3 $R/bench.py --steps 100 --warmup 10 --core-only > $R/gpurun_out/pmc_sq$i.log 2>&1
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_sqw$i -- python3 $R/tools/stack_time.py 4096 > $R/gpurun_out/pmc_sqw$i.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, json
# (bench.py --core-only launches two instantiations: <0, .> in the raw loop, <5, .> -- the rotated kernel -- in the GenesisEnv.step loop)
for pat, kern, dst in (("gpurun_out/pmc_sq[0-9]*/*/*counter_collection.csv", "mir_step_kernel<0", "gpurun_out/sq_counters.json"),
                       ("gpurun_out/pmc_sq[0-9]*/*/*counter_collection.csv", "mir_step_kernel<5", "gpurun_out/sq_counters_rotated.json"),
                       ("gpurun_out/pmc_sqw*/*/*counter_collection.csv", "mir_step64_kernel", "gpurun_out/sq_counters_step64.json")):
    out = {}
    for f in glob.glob(pat):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            v = sorted(v.values())
            out[k] = {"launches": len(v), "median_per_launch": v[len(v) // 2]}
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
PY
rm -rf gpurun_out/pmc_sq*/ gpurun_out/pmc_sqw*/
