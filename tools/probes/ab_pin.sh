run() { python3 bench.py --gpus 1 --steps 20 --warmup 5 --core-only 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read()); r = d['repeat_us_per_step']
print(sys.argv[1], ': value', round(d['value'] / 1e6, 1), 'M | median region', round(r['median'], 2), '| p95', round(r['p95'], 2), '| max', round(r['max'], 1))" $1; }
for i in 1 2 3 4 5; do MIR_BENCH_PIN=0 run free; MIR_BENCH_PIN=1 run pinned; done
