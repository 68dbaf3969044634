run() {
  python3 bench.py --gpus 1 --steps 2000 --warmup 50 --core-only 2>&1 | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print(sys.argv[1], round(d['value'] / 1e6, 2), 'M | median region', round(d.get('value_median_region', 0) / 1e6, 2), 'M |', round(d['ms_per_step'] * 1e3, 3), 'us/step | hot', round(d['hot_path_rate'] / 1e6, 2), 'M')" $1
}
for r in 1 2 3; do
  MIR_NO_EARLY_MASK=1 MIR_TERM_DENSE=1 run late+dense
  MIR_TERM_DENSE=1 run early+dense
  MIR_NO_EARLY_MASK=1 run late+lines
  run early+lines
done
