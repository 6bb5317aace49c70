#!/bin/bash
# SQ counters of the wave-specialised adjoint at the benchmark size (run on the GPU box through gpurun):  tools/pmc_adjoint.sh [B]
# writes gpurun_out/pmc_adj/...; prints per-dispatch means for solve_bwd_ws_kernel (one rocprofv3 --pmc pass per counter group)
set -e
B=${1:-4096}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_adj
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
export HODE_NO_BUILD=1
cd /tmp
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $C --output-format csv -d $OUT/$N -- python3 $R/tools/time_adjoint.py $B > $OUT/log_$N.txt 2>&1 || echo "pmc $C failed"
done
python3 - <<PY
import collections, csv, glob
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "solve_bwd_ws" in k:
            acc[(k.split("(")[0][-40:], row["Counter_Name"])].append(float(row["Counter_Value"]))
d = {}
for (k, name), v in sorted(acc.items()):
    print(f"{k:42s} {name:24s} n={len(v):3d} mean {sum(v)/len(v):16.1f}")
    d.setdefault(k, {})[name] = sum(v) / len(v)
for k, c in d.items():
    if "SQ_WAVE_CYCLES" in c and "SQ_ACTIVE_INST_VALU" in c:
        print(k, "VALU busy (x4 waves / wave cycles):", round(4 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], 3),
              " cycles per VALU instruction:", round(4 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"], 2),
              " VALU instructions per wave:", round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"]))
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in d.get(k, {}):
        print(k, "SQ_WAIT_ANY / SQ_WAVE_CYCLES:", round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3))
PY
