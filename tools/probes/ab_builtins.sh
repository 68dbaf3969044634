run() { python3 bench.py --gpus 1 --steps 2000 --warmup 50 --core-only 2>&1 | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print(sys.argv[1], round(d['value'] / 1e6, 2), 'M |', round(d['ms_per_step'] * 1e3, 3), 'us/step | kernel', round(d['roofline']['kernel_us'], 3), 'us')" $1; }
F=gym-genesis_amd/gym_genesis/backend/_mirfast.so
for r in 1 2 3; do
  run builtin
  mv $F /tmp/_mirfast.keep; run ctypes; mv /tmp/_mirfast.keep $F
done
