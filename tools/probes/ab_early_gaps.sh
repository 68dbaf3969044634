set -u
R=$(pwd); export TMPDIR=/tmp; cd /tmp
for mode in "MIR_SYNC_MODE=3" "MIR_SYNC_MODE=3 MIR_NO_EARLY_MASK=1" "MIR_SYNC_MODE=1" "MIR_SYNC_MODE=0" "MIR_SYNC_MODE=2"; do
  rm -rf /tmp/gaps_x
  env $mode rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps_x -- python3 $R/tools/probes/api_gaps.py run > /dev/null 2>&1
  echo "== $mode"; python3 $R/tools/probes/api_gaps.py parse /tmp/gaps_x
done
