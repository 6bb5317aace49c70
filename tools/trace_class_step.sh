#!/bin/bash
# kernel trace of the class-path optimisation step at the reference's batch (run on the GPU box): which kernels the ~2 ms are
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_class
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp HODE_NO_BUILD=1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/time_class_step.py ${1:-32} > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time total %.1f ms over the run (55 steps)" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%7.1f us/step  calls/step %5.1f  avg %7.1f us  %s" % (float(r["TotalDurationNs"]) / 55e3, float(r["Calls"]) / 55, float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
