#!/bin/bash
# SQ counters of the product forward kernels at the benchmark size (GPU box, through gpurun):  tools/pmc_fwd.sh [B]
# writes gpurun_out/pmc_fwd_cur/...; summarise with `python tools/pmc_fwd_variant.py cur`.  Counter passes only (no trace domains).
set -e
B=${1:-4096}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_fwd_cur
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $C --output-format csv -d $OUT/$N -- python3 $R/tools/time_fwd_tape.py $B > $OUT/log_$N.txt 2>&1 || echo "pmc $C failed"
done
echo done
