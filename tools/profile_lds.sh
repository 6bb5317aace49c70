#!/bin/bash
# GPU box: LDS counters of the solve kernels (own --pmc pass, no tracing domains).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_lds
rm -rf $OUT && mkdir -p $OUT
# build OUTSIDE the profiler: under rocprofv3 the tool library has initialised the GPU before Python starts, and bench.py must
# not spawn make from there (it gets --no-build and fails loudly on a missing or stale artefact instead) -- as tools/profile_gpu.sh
make -C $R/hybrid-ode-for-glp-1-and-glucose_amd/csrc -j8 > $OUT/build.log 2>&1
make -C $R/oracle -s >> $OUT/build.log 2>&1
export TMPDIR=/tmp
export HODE_NO_BUILD=1
cd /tmp
ARGS="--steps 3 --warmup 1 --train-steps 2 --no-cpu-baseline --no-vi --no-generic --no-data-side --no-sobol --no-class-path --no-build"
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $R/bench.py $ARGS > $OUT/log_$N.txt 2>&1 || echo "pmc $C failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "hode::solve" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
