#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: kernel-trace stats of the bench command, the two HBM-traffic PMC passes over
# the raw launch loop, and the same two passes over the calibration copy.  Summaries land in gpurun_out/ and are then copied into
# profiles/<round>/.  Counters are collected in their own runs (--pmc only, never together with a trace).
set -u
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
# (1) the bench command itself: API loop + raw loop, both launch mir_step_kernel<0>
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- python3 $R/bench.py --steps 1000 --warmup 50 --core-only > $R/gpurun_out/bench_under_rocprof.log 2>&1
# (2) only raw back-to-back launches: the kernel's own duration
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace_raw -- python3 $R/bench.py --steps 1000 --warmup 50 --core-only --raw-only > $R/gpurun_out/bench_raw_under_rocprof.log 2>&1
# (3) HBM traffic of the step kernel, one counter per pass
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py --steps 200 --warmup 10 --min-time 0 --core-only --raw-only > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/bench.py --steps 200 --warmup 10 --min-time 0 --core-only --raw-only > $R/gpurun_out/pmc_write.log 2>&1
# (4) the same two counters on a copy of known size with the same access width
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_cfetch -- python3 $R/tools/traffic_calib.py > $R/gpurun_out/pmc_cfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_cwrite -- python3 $R/tools/traffic_calib.py > $R/gpurun_out/pmc_cwrite.log 2>&1
# (5) the same two counters over the API loop: its launches are the rotated kernel (second half of this step + first half of the next)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_afetch -- python3 $R/bench.py --steps 200 --warmup 10 --min-time 0 --core-only > $R/gpurun_out/pmc_afetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_awrite -- python3 $R/bench.py --steps 200 --warmup 10 --min-time 0 --core-only > $R/gpurun_out/pmc_awrite.log 2>&1
cd $R
python3 tools/summarise_pmc.py gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_cfetch gpurun_out/prof_cwrite > gpurun_out/pmc_hbm_traffic.json 2> gpurun_out/pmc_summary.err
MIR_PMC_KERNEL="mir_step_kernel<5" python3 tools/summarise_pmc.py gpurun_out/prof_afetch gpurun_out/prof_awrite gpurun_out/prof_cfetch gpurun_out/prof_cwrite > gpurun_out/pmc_hbm_traffic_api.json 2>> gpurun_out/pmc_summary.err
find gpurun_out/prof_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/bench_kernel_stats.csv \;
find gpurun_out/prof_trace_raw -name "*kernel_stats.csv" -exec cp {} gpurun_out/bench_raw_kernel_stats.csv \;
# raw counter dumps are large: keep only the summaries
rm -rf gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_cfetch gpurun_out/prof_cwrite gpurun_out/prof_afetch gpurun_out/prof_awrite
find gpurun_out/prof_trace gpurun_out/prof_trace_raw -name "*kernel_trace.csv" -delete
cat gpurun_out/pmc_hbm_traffic.json
head -5 gpurun_out/bench_kernel_stats.csv gpurun_out/bench_raw_kernel_stats.csv
tail -c 1500 gpurun_out/bench_raw_under_rocprof.log
