#!/bin/bash
# on the GPU box, from the repo root: kernel trace of tools/probes/early_trace.py, then the overlap report
# usage: early_trace.sh [N] [timeline]   (MODE=expert in the environment: the reference's expert)
set -u
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_early -- python3 $R/tools/probes/early_trace.py > $R/gpurun_out/early_trace_run.log 2>&1
cd $R
T=$(find gpurun_out/prof_early -name "*kernel_trace.csv" | head -1)
python3 tools/probes/early_trace_report.py $T ${1:-60} ${2:-} > gpurun_out/early_trace_report.log 2>&1
rm -rf gpurun_out/prof_early
tail -3 gpurun_out/early_trace_run.log
cat gpurun_out/early_trace_report.log
