#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: kernel-trace stats and the two HBM-traffic PMC passes
# for the bench command.  Outputs under gpurun_out/prof_*; summaries are then copied into profiles/<round>/.
set -u
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- python3 $R/bench.py --steps 1000 --warmup 50 --core-only > $R/gpurun_out/bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py --steps 200 --warmup 10 --core-only > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/bench.py --steps 200 --warmup 10 --core-only > $R/gpurun_out/pmc_write.log 2>&1
cd $R
python3 tools/summarise_pmc.py gpurun_out/prof_fetch gpurun_out/prof_write > gpurun_out/pmc_hbm_traffic.json 2> gpurun_out/pmc_summary.err
find gpurun_out/prof_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/bench_kernel_stats.csv \;
# raw counter dumps are large: keep only the summaries
rm -rf gpurun_out/prof_fetch gpurun_out/prof_write
find gpurun_out/prof_trace -name "*kernel_trace.csv" -delete
