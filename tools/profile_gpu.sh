#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + stats, then PMC passes (own runs, no tracing domains mixed in).
# usage: tools/profile_gpu.sh <tag>      outputs under gpurun_out/prof_<tag>/
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 5 --warmup 2 --train-steps 2 --no-cpu-baseline --no-vi"
# the kernel sources these counters belong to (bench.py reports PMC traffic only while this hash matches)
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.kernel_source_sha())" > $OUT/kernel_source_sha.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.log 2>&1
echo "trace done"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_$N.log 2>&1 || echo "pmc $C failed"
  echo "pmc $C done"
done
find $OUT -name "*.csv" | head -50
