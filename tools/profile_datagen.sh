#!/bin/bash
# Run on the GPU box: HBM traffic of the data-side kernels (separate --pmc passes, no tracing domains).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_datagen
rm -rf $OUT && mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/tools/time_datagen.py --cpu-sample 0 --reps 2 > $OUT/log_$C.txt 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "hode::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    fe = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]); wr = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    print(f"{k}: fetch {2 * fe * 1024 / 1e6:.1f} MB (2 x FETCH_SIZE)  write {wr * 1024 / 1e6:.1f} MB")
PY
