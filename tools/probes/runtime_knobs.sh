run() { python3 bench.py --gpus 1 --steps 2000 --warmup 50 --core-only 2>&1 | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print(sys.argv[1], round(d['value'] / 1e6, 2), 'M |', round(d['ms_per_step'] * 1e3, 3), 'us/step | kernel', round(d['roofline']['kernel_us'], 3), 'us | fused', round(d['roofline_fused_launch']['kernel_us'], 3))" "$1"; }
run default
HIP_FORCE_DEV_KERNARG=1 run devkernarg1
HIP_FORCE_DEV_KERNARG=0 run devkernarg0
run default
GPU_MAX_HW_QUEUES=1 run hwq1
HSA_ENABLE_INTERRUPT=0 run nointerrupt
HIP_FORCE_DEV_KERNARG=1 run devkernarg1
run default
