"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per mir_step_kernel launch, corrected with the
calibration passes over tools/traffic_calib.py (a copy of known size with the step kernel's access shape).

    python tools/summarise_pmc.py <fetch_dir> <write_dir> [<calib_fetch_dir> <calib_write_dir>]
"""
import csv
import glob
import json
import os
import sys

ALGO_BYTES = float(os.environ.get("MIR_PMC_ALGO_BYTES", 489.0 * 4096))  # (1109 x 4096 for the stack kernel)
CALIB_BYTES = 16 * 1024 * 1024 * 4  # tools/traffic_calib.py: 64 MiB read and 64 MiB written per launch


def per_launch(d, counter, kernel):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                rows.append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
    # one row per (dispatch, xcd/instance) in some rocprofv3 versions: sum per dispatch
    acc = {}
    for k, v in rows:
        acc[k] = acc.get(k, 0.0) + v
    vals = sorted(acc.values())
    if not vals:
        return {"launches": 0, "mean_KB_per_launch": None, "median_KB_per_launch": None, "min": None, "max": None}
    return {"launches": len(vals), "mean_KB_per_launch": sum(vals) / len(vals), "median_KB_per_launch": vals[len(vals) // 2],
            "min": vals[0], "max": vals[-1]}


def main():
    kname = os.environ.get("MIR_PMC_KERNEL", "mir_step_kernel<0")  # e.g. "mir_step_kernel<5" = the rotated launch of the API path
    fetch = per_launch(sys.argv[1], "FETCH_SIZE", kname)
    write = per_launch(sys.argv[2], "WRITE_SIZE", kname)
    out = {"FETCH_SIZE": fetch, "WRITE_SIZE": write, "kernel": kname, "grid": "1024 workgroups x 64 threads (B=4096)",
           "algorithmic_bytes_per_launch": ALGO_BYTES}
    raw = 1024.0 * (fetch["median_KB_per_launch"] + write["median_KB_per_launch"])
    out["hbm_bytes_per_launch_uncorrected"] = raw
    kf = kw = None
    if len(sys.argv) >= 5:
        cf = per_launch(sys.argv[3], "FETCH_SIZE", "k_copy_rows")
        cw = per_launch(sys.argv[4], "WRITE_SIZE", "k_copy_rows")
        out["calibration"] = {"kernel": "k_copy_rows (mir_debug_copy_rows): 4 B per lane, 64-thread workgroups, 64 MiB in + 64 MiB out per launch",
                              "known_bytes_each_way": CALIB_BYTES, "FETCH_SIZE": cf, "WRITE_SIZE": cw}
        if cf["launches"] and cw["launches"]:
            kf = CALIB_BYTES / (1024.0 * cf["median_KB_per_launch"])
            kw = CALIB_BYTES / (1024.0 * cw["median_KB_per_launch"])
            out["calibration"]["fetch_factor"] = kf
            out["calibration"]["write_factor"] = kw
    if kf is not None:
        corrected = 1024.0 * (kf * fetch["median_KB_per_launch"] + kw * write["median_KB_per_launch"])
        out["hbm_bytes_per_launch"] = corrected
        loop = "`bench.py --core-only` (the GenesisEnv.step loop)" if os.environ.get("MIR_PMC_KERNEL") else "`bench.py --core-only --raw-only`"
        out["note"] = (f"separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over {loop}; per-launch medians in the "
                       "KB units rocprofv3 reports, each multiplied by the factor that makes the same counter read 64 MiB on the "
                       "calibration copy, which has this kernel's access width (4 B per lane)")
    else:
        out["hbm_bytes_per_launch"] = raw
        out["note"] = "separate --pmc passes; no calibration pass found, counters left as rocprofv3 reports them"
    out["ratio_to_algorithmic"] = out["hbm_bytes_per_launch"] / ALGO_BYTES
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
