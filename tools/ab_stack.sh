#!/bin/bash
# Same-box A/B of two builds of libmirigid.so on the stack tasks (run on the GPU box from the repo root): alternates the two libraries
# three times under tools/stack_time.py (box-to-box differences are +-2 %, larger than most single changes to the wave kernel).
# Usage: bash tools/ab_stack.sh <a.so> <b.so>
set -u
A=$1; B=$2
L=gym-genesis_amd/csrc/libmirigid.so
cp $L /tmp/ab_keep.so
for r in 1 2 3; do
  cp $A $L; echo "A $(python3 tools/stack_time.py 2>&1 | grep -E 'franka|so101' | awk '{print $4}' | tr '\n' ' ')"
  cp $B $L; echo "B $(python3 tools/stack_time.py 2>&1 | grep -E 'franka|so101' | awk '{print $4}' | tr '\n' ' ')"
done
cp /tmp/ab_keep.so $L
