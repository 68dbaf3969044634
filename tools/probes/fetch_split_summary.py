import csv, glob, sys, os
for d, counter in ((sys.argv[1], "FETCH_SIZE"), (sys.argv[2], "WRITE_SIZE")):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "mir_step_kernel" in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                k = int(r["Dispatch_Id"]); acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
    ids = sorted(acc)
    v = [acc[i] for i in ids]
    print(counter, "launches", len(v))
    for name, sl in (("all outputs", slice(10, 50)), ("physics only", slice(50, 90)), ("packed rows", slice(90, 130))):
        w = sorted(v[sl]); print(f"  {name:14s} median {w[len(w)//2]:9.1f} KB  min {w[0]:9.1f} max {w[-1]:9.1f}")
