#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: per-kernel stats of the pixels path (tools/render_time.py) and the
# HBM write-traffic PMC pass.  Summaries land in gpurun_out/ and are copied into profiles/<round>/ by hand.
set -u
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_render -- python3 $R/tools/render_time.py 1024 > $R/gpurun_out/render_under_rocprof.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_render_write -- python3 $R/tools/render_time.py 1024 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_render_fetch -- python3 $R/tools/render_time.py 1024 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/prof_render_sq -- python3 $R/tools/render_time.py 1024 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/prof_render_sq2 -- python3 $R/tools/render_time.py 1024 > /dev/null 2>&1
cd $R
find gpurun_out/prof_render -name "*kernel_stats.csv" -exec cp {} gpurun_out/render_kernel_stats.csv \;
python3 - <<'PY' > gpurun_out/render_pmc.json
import csv, glob, json, os
out = {}
for d in ("prof_render_write", "prof_render_fetch", "prof_render_sq", "prof_render_sq2"):
    for f in glob.glob(os.path.join("gpurun_out", d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            kn = r.get("Kernel_Name", "")
            if "mir_render_" not in kn and "fill" not in kn.lower():
                continue
            grid = ("render " if "mir_render_" in kn else "fill ") + r.get("Grid_Size", "")
            key = (r["Counter_Name"], grid)
            acc.setdefault(key, {}).setdefault(int(r.get("Dispatch_Id", 0)), 0.0)
            acc[key][int(r.get("Dispatch_Id", 0))] += float(r["Counter_Value"])
        for (name, grid), v in acc.items():
            vals = sorted(v.values())
            out.setdefault(f"grid={grid}", {})[name] = {"launches": len(vals), "median": vals[len(vals) // 2]}
# HBM bytes per full-resolution render launch (largest render grid): KB as rocprofv3 reports them, FETCH_SIZE uncorrected
big = max((k for k in out if k.startswith("grid=render")), key=lambda k: int(k.split()[-1]), default=None)
if big and "WRITE_SIZE" in out[big] and "FETCH_SIZE" in out[big]:
    out["hbm_bytes_per_launch"] = 1024.0 * (out[big]["WRITE_SIZE"]["median"] + out[big]["FETCH_SIZE"]["median"])
    out["algorithmic_bytes_per_launch"] = 1024 * 480 * 640 * 3
print(json.dumps(out, indent=1))
PY
rm -rf gpurun_out/prof_render_write gpurun_out/prof_render_fetch gpurun_out/prof_render_sq gpurun_out/prof_render_sq2
find gpurun_out/prof_render -name "*kernel_trace.csv" -delete
