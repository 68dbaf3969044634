#!/bin/bash
# Run on the GPU box from the repo root: kernel-trace stats of the scripted grasp through GenesisEnv.step with and without exact
# contacts (tools/exact_time.py): the list-mode launches of the wave kernel (mir_step64_kernel<0, true>) and of the 16-lane kernel's
# first half (mir_step_kernel<3, 5>) beside the rotated launches.  Summary -> gpurun_out/exact_kernel_stats.csv.
set -u
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_exact -- python3 $R/tools/exact_time.py > $R/gpurun_out/exact_under_rocprof.log 2>&1
cd $R
find gpurun_out/prof_exact -name "*kernel_stats.csv" -exec cp {} gpurun_out/exact_kernel_stats.csv \;
find gpurun_out/prof_exact -name "*kernel_trace.csv" -delete
python3 tools/exact_time.py > gpurun_out/exact_time.log 2>&1
head -12 gpurun_out/exact_kernel_stats.csv
