#!/bin/bash
# Same-box comparison of N builds of libmirigid.so on tools/stack_time.py (CubeStack step time, Franka / SO-101).  Usage: bash tools/probes/stack_abn.sh a.so b.so ...
set -u
L=gym-genesis_amd/csrc/libmirigid.so
cp $L /tmp/stack_keep.so
for i in 1 2 3; do for f in "$@"; do cp $f $L; echo "$(basename $f) $(python3 tools/stack_time.py 2>&1 | grep -E 'franka|so101' | awk '{print $4}' | tr '\n' ' ')"; done; done
cp /tmp/stack_keep.so $L
