#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: per-kernel stats of the stack tasks on the wave-per-env kernel.
set -u
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stack -- python3 $R/tools/stack_time.py 4096 > $R/gpurun_out/stack_under_rocprof.log 2>&1
# HBM traffic of the wave kernel, one counter per pass (+ the calibration copy, as in collect_profiles.sh)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_sfetch -- python3 $R/tools/stack_time.py 4096 > $R/gpurun_out/pmc_sfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_swrite -- python3 $R/tools/stack_time.py 4096 > $R/gpurun_out/pmc_swrite.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_cfetch2 -- python3 $R/tools/traffic_calib.py > $R/gpurun_out/pmc_cfetch2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_cwrite2 -- python3 $R/tools/traffic_calib.py > $R/gpurun_out/pmc_cwrite2.log 2>&1
cd $R
MIR_PMC_KERNEL="mir_step64_kernel" MIR_PMC_ALGO_BYTES=4542464 python3 tools/summarise_pmc.py gpurun_out/prof_sfetch gpurun_out/prof_swrite gpurun_out/prof_cfetch2 gpurun_out/prof_cwrite2 > gpurun_out/pmc_hbm_traffic_step64.json 2> gpurun_out/pmc_summary64.err
rm -rf gpurun_out/prof_sfetch gpurun_out/prof_swrite gpurun_out/prof_cfetch2 gpurun_out/prof_cwrite2
find gpurun_out/prof_stack -name "*kernel_stats.csv" -exec cp {} gpurun_out/stack_kernel_stats.csv \;
find gpurun_out/prof_stack -name "*kernel_trace.csv" -delete
python3 tools/phase_profile64.py 64 > gpurun_out/stack_phase_profile.txt 2>&1
