set -u
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh > gpurun_out/collect_profiles.log 2>&1
bash tools/collect_sq_counters.sh > gpurun_out/collect_sq.log 2>&1
bash tools/profile_stack.sh > gpurun_out/profile_stack.log 2>&1
bash tools/profile_render.sh > gpurun_out/profile_render.log 2>&1
bash tools/profile_exact.sh > gpurun_out/profile_exact.log 2>&1
bash tools/profile_ik.sh > gpurun_out/profile_ik.log 2>&1
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.out 2> gpurun_out/bench_driver.err
ls gpurun_out | head -50
