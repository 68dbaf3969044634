#!/bin/bash
# Same-box comparison of N builds of libmirigid.so on bench.py's pixels leg.  Usage (GPU box, repo root): bash tools/probes/pixels_abn.sh a.so b.so ...
set -u
L=gym-genesis_amd/csrc/libmirigid.so
cp $L /tmp/pix_keep.so
run() {
  python3 -c "
import sys; sys.argv=['bench.py']
import bench, torch
r = bench.pixels_bench(torch, torch.device('cuda', 0))
print('$1', round(r['us_per_render'], 1), 'us per render', round(r['roofline']['frac'], 3), '| fill', round(r['fill_us_same_buffer'], 1), '| host', round(r['host_enqueue_us_per_render'], 1))" 2>/dev/null | tail -1
}
for i in 1 2 3; do for f in "$@"; do cp $f $L; run $(basename $f); done; done
cp /tmp/pix_keep.so $L
