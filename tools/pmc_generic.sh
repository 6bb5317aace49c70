#!/bin/bash
# SQ counters of the generic-path kernels (run on the GPU box through gpurun):  tools/pmc_generic.sh H L T B
# writes gpurun_out/pmc_gen/...; prints per-dispatch means for the generic forward / adjoint kernels
set -e
H=${1:-128}; L=${2:-5}; T=${3:-61}; B=${4:-1024}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_gen
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_GDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $C --output-format csv -d $OUT/$N -- python3 $R/tools/time_generic_b.py $H $L $T $B > $OUT/log_$N.txt 2>&1 || echo "pmc $C failed"
done
python3 - <<PY
import collections, csv, glob
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "generic" in k:
            kind = "bwd" if "solve_bwd" in k else ("fwd_tape" if ("Lb1ELb" in k or ", true, " in k) else "fwd")
            acc[(kind, row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, name), v in sorted(acc.items()):
    print(f"{k:9s} {name:24s} n={len(v):3d} mean {sum(v)/len(v):16.1f}")
PY
