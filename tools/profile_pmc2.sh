#!/bin/bash
# extra PMC groups for the adjoint kernel (own runs, no tracing domains)
set -e
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
# build OUTSIDE the profiler: under rocprofv3 the tool library has initialised the GPU before Python starts, and bench.py must
# not spawn make from there (it gets --no-build and fails loudly on a missing or stale artefact instead) -- as tools/profile_gpu.sh
make -C $R/hybrid-ode-for-glp-1-and-glucose_amd/csrc -j8 > $OUT/build.log 2>&1
make -C $R/oracle -s >> $OUT/build.log 2>&1
export TMPDIR=/tmp
export HODE_NO_BUILD=1
cd /tmp
ARGS="--steps 3 --warmup 1 --train-steps 2 --no-cpu-baseline --no-sobol --no-class-path --no-build"
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$i -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_$i.log 2>&1 || echo "pmc $C failed"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$OUT/pmc_*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        if 'solve_' in k:
            print(k, {c: round(sum(v)/len(v)/1e6, 1) for c, v in d.items()}, '(millions)')
PY
