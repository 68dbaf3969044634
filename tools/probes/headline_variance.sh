for i in 1 2 3 4 5 6 7 8; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --core-only 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read()); r = d['repeat_us_per_step']
print('run', sys.argv[1], ': value', round(d['value'] / 1e6, 1), 'M | mean', round(d['ms_per_step'] * 1e3, 2), 'us/step | median region', round(r['median'], 2), '| p95', round(r['p95'], 2), '| max', round(r['max'], 1), '| hot path', round(d['hot_path_rate'] / 1e6, 1), 'M')" $i; done
