#!/bin/bash
# Same-box A/B of two builds of libmirigid.so on the LDS bank-conflict counters of the step kernels (one --pmc pass each) and on the
# kernel times (tools/ab_bench.sh).  Usage (GPU box, repo root): bash tools/lds_conflict_ab.sh <a.so> <b.so>
set -u
R=$(pwd); export TMPDIR=/tmp
L=gym-genesis_amd/csrc/libmirigid.so
cp $L /tmp/lds_keep.so
for v in A B; do
  if [ $v = A ]; then cp $1 $L; else cp $2 $L; fi
  (cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_lds_$v -- python3 $R/bench.py --steps 100 --warmup 10 --core-only > $R/gpurun_out/pmc_lds_$v.log 2>&1)
  python3 - $v <<'PY'
import csv, glob, collections, sys
v = sys.argv[1]
for kern in ("mir_step_kernel<0", "mir_step_kernel<5"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"gpurun_out/pmc_lds_{v}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    med = {k: sorted(x.values())[len(x) // 2] for k, x in acc.items()}
    if med:
        print(v, kern, {k: int(x) for k, x in med.items()}, "conflict / active-lds = %.3f" % (med.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, med.get("SQ_ACTIVE_INST_LDS", 1))))
PY
  rm -rf gpurun_out/pmc_lds_$v
done
cp /tmp/lds_keep.so $L
bash tools/ab_bench.sh $1 $2
