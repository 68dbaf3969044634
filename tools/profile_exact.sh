#!/bin/bash
# Run on the GPU box from the repo root: kernel-trace stats of the scripted grasp through GenesisEnv.step with and without exact
# contacts (tools/exact_time.py), then of the reference's expert (tools/expert_time.py): the launches of the three-contacts-per-lane
# instantiation -- mir_step_kernel<6, 5, 3> on the list of deferred envs, mir_step_kernel<7, 5, 3> on the whole batch in a heavy phase --
# beside the rotated launches.  Summaries -> gpurun_out/exact_kernel_stats.csv, gpurun_out/expert_kernel_stats.csv, *.log.
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
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_expert -- python3 $R/tools/expert_time.py 1 > $R/gpurun_out/expert_under_rocprof.log 2>&1
cd $R
find gpurun_out/prof_expert -name "*kernel_stats.csv" -exec cp {} gpurun_out/expert_kernel_stats.csv \;
rm -rf gpurun_out/prof_expert gpurun_out/prof_exact
python3 tools/expert_time.py 2 > gpurun_out/expert_time.log 2>&1
head -8 gpurun_out/expert_kernel_stats.csv
head -12 gpurun_out/exact_kernel_stats.csv
