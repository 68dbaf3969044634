#!/bin/bash
# Same-box comparison of N builds of libmirigid.so with tools/render_time.py (home pose + floor alone).  Usage: bash tools/probes/render_abn.sh a.so b.so ...
set -u
L=gym-genesis_amd/csrc/libmirigid.so
cp $L /tmp/pix_keep.so
for i in 1 2 3; do for f in "$@"; do cp $f $L; echo "$(basename $f): $(python3 tools/render_time.py 1024 floor 2>/dev/null | head -2 | tr '\n' ' ')"; done; done
cp /tmp/pix_keep.so $L
