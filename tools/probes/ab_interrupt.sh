# the driver's bench command (20-step regions) with the HSA runtime's waits on interrupts (default) and busy-waiting, alternating
run() { python3 bench.py --gpus 1 --steps 20 --warmup 5 --core-only 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print(sys.argv[1], round(d['value'] / 1e6, 2), 'M | median region', round(d['value_median_region'] / 1e6, 2), 'M |', round(d['ms_per_step'] * 1e3, 3), 'us/step | hot', round(d['hot_path_rate'] / 1e6, 2))" "$1"; }
for r in 1 2 3; do
  run default
  HSA_ENABLE_INTERRUPT=0 run nointerrupt
done
