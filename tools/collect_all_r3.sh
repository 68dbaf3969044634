set -u
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh > gpurun_out/collect_profiles.log 2>&1
bash tools/collect_sq_counters.sh > gpurun_out/collect_sq.log 2>&1
bash tools/profile_stack.sh > gpurun_out/profile_stack.log 2>&1
ls gpurun_out | head -50
