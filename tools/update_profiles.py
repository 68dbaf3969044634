"""Copy the summaries that tools/collect_profiles.sh and tools/gpu_check.sh left in gpurun_out/ into profiles/<round>/ (tracked),
trimming the kilobyte-long template names torch's kernels have in the rocprofv3 CSVs."""
import csv
import os
import re
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r3"
dst = os.path.join(R, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
src = os.path.join(R, "gpurun_out")
for name in ("pmc_hbm_traffic.json", "pmc_hbm_traffic_api.json", "pmc_hbm_traffic_step64.json", "sq_counters.json", "sq_counters_rotated.json",
             "sq_counters_step64.json", "stack_kernel_stats.csv", "stack_phase_profile.txt", "render_pmc.json"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, name))
for name in ("bench_kernel_stats.csv", "bench_raw_kernel_stats.csv", "render_kernel_stats.csv"):
    if not os.path.exists(os.path.join(src, name)):
        continue
    rows = list(csv.reader(open(os.path.join(src, name))))
    for r in rows:
        if len(r[0]) > 200:
            r[0] = r[0][:120] + "...(name truncated)"
    csv.writer(open(os.path.join(dst, name), "w", newline="")).writerows(rows)
for log, out in (("bench_under_rocprof.log", "bench_under_rocprof.log"), ("bench_raw_under_rocprof.log", "bench_raw_under_rocprof.log"),
                 ("bench_driver.out", "bench_driver_command.log")):
    if not os.path.exists(os.path.join(src, log)):
        continue
    text = open(os.path.join(src, log)).read()
    m = re.findall(r'^\{"metric".*$', text, flags=re.M)
    open(os.path.join(dst, out), "w").write((m[-1] if m else text[-4000:]) + "\n")
for k in range(4):
    p = os.path.join(src, f"bench_sync{k}.out")
    if os.path.exists(p):
        m = re.findall(r'^\{"metric".*$', open(p).read(), flags=re.M)
        if m:
            open(os.path.join(dst, f"bench_core_only_sync_mode{k}.log"), "w").write(m[-1] + "\n")
print("profiles updated:", sorted(os.listdir(dst)))
