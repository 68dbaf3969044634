"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per mir_step_kernel launch."""
import csv
import glob
import json
import os
import sys


def per_launch(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "mir_step_kernel" in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                rows.append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
    # one row per (dispatch, xcd/instance) in some rocprofv3 versions: sum per dispatch
    acc = {}
    for k, v in rows:
        acc[k] = acc.get(k, 0.0) + v
    vals = list(acc.values())
    return {"launches": len(vals), "mean_KB_per_launch": sum(vals) / max(len(vals), 1),
            "min": min(vals) if vals else None, "max": max(vals) if vals else None}


def main():
    fetch = per_launch(sys.argv[1], "FETCH_SIZE")
    write = per_launch(sys.argv[2], "WRITE_SIZE")
    out = {"FETCH_SIZE": fetch, "WRITE_SIZE": write, "kernel": "mir_step_kernel",
           "grid": "1024 workgroups x 64 threads (B=4096)",
           "hbm_bytes_per_launch_uncorrected": 1024.0 * (fetch["mean_KB_per_launch"] + write["mean_KB_per_launch"]),
           "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over `bench.py --steps 200`; KB units as rocprofv3 reports "
                   "them; the guide's x2 FETCH_SIZE correction applies to 16 B/lane streaming reads only (this kernel reads "
                   "4 B/lane rows), so the figures are left uncorrected"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
