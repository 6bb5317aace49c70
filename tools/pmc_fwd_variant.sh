#!/bin/bash
# SQ counters of one forward-kernel variant (run on the GPU box through gpurun):  tools/pmc_fwd_variant.sh <regs|wg|quad|rows> [B]
# writes gpurun_out/pmc_fwd_<mode>/...; summarise with tools/pmc_fwd_variant.py
set -e
MODE=${1:-regs}; B=${2:-8192}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_fwd_$MODE
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp HODE_FWD=$MODE
cd /tmp
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $C --output-format csv -d $OUT/$N -- python3 $R/tools/fwd_variants.py $B > $OUT/log_$N.txt 2>&1 || echo "pmc $C failed"
done
echo done $MODE
