"""Reads a rocprofv3 kernel trace csv: for every launch of the list instantiation <6, ., 3> prints when it started / ended relative to
the main launch <5, ., 1> nearest in time, and the period between consecutive main launches; with a third argument "timeline": every
kernel of the last N list launches' neighbourhood, start relative to the first and duration."""
import bisect, csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
main = [(s, e) for s, e, n, q in rows if "mir_step_kernel<5" in n]
lst = [(s, e) for s, e, n, q in rows if "mir_step_kernel<6" in n or "mir_step_kernel<9" in n]
print(f"{len(main)} main launches, {len(lst)} list launches")
if len(sys.argv) > 3 and sys.argv[3] == "timeline":
    first = lst[-N][0] - 200000 if len(lst) >= N else rows[0][0]
    t0 = None
    for s, e, n, q in rows:
        if s < first:
            continue
        if t0 is None:
            t0 = s
        import re
        mm = re.search(r"(mir_\w+(<[^>]*>)?|k_\w+)", n)
        short = mm.group(1) if mm else n[:48]
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:6.1f}  q{q}  {short}")
        if (s - t0) > 3e6:
            break
    sys.exit(0)
ms = [s for s, _ in main]
out = []
for s, e in lst[-N:]:
    i = max(bisect.bisect_right(ms, s + 30000) - 1, 0)   # the main launch that started before (or up to 30 us after) the list launch
    m0, m1 = main[i]
    nxt = main[i + 1][0] if i + 1 < len(main) else 0
    out.append(f"list start {(s - m0) / 1e3:+7.1f} us after main start, lasts {(e - s) / 1e3:6.1f}; main lasts {(m1 - m0) / 1e3:5.1f}; next main starts {(nxt - m0) / 1e3:6.1f} after this one")
print("\n".join(out))
