#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: per-kernel stats of the stack tasks on the wave-per-env kernel.
set -u
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stack -- python3 $R/tools/stack_time.py 4096 > $R/gpurun_out/stack_under_rocprof.log 2>&1
cd $R
find gpurun_out/prof_stack -name "*kernel_stats.csv" -exec cp {} gpurun_out/stack_kernel_stats.csv \;
find gpurun_out/prof_stack -name "*kernel_trace.csv" -delete
python3 tools/phase_profile64.py 64 > gpurun_out/stack_phase_profile.txt 2>&1
