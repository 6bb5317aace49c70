#!/bin/bash
# Build container side: refresh everything under profiles/ that depends on the kernels, in the one order that works
# (the PMC stamp must be written BEFORE bench.py runs, and stale files under gpurun_out/ must not be averaged in), then DESIGN.md.
#   tools/refresh_profiles.sh [tag]        e.g.  tools/refresh_profiles.sh r03
set -e
TAG=${1:-r03}
cd "$(dirname "$0")/.."
GPURUN=/usr/local/graft/bin/gpurun
rm -rf gpurun_out/prof_$TAG
# ONE call = ONE box: the GPU suite, trace + PMC passes, their summary (written on the box too: bench.py reads profiles/pmc_traffic.json),
# then the plain bench.py line -- so the rocprofv3 averages and the bench's HIP-event figures are from the same machine (boxes differ by
# 2-3 %, and by more in their first launches).  The summary is made again here, from the merged gpurun_out/, for the tracked copies.
$GPURUN --timeout 1200 -- "timeout -k 10 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/${TAG}_gputest.log 2>&1; tail -1 gpurun_out/${TAG}_gputest.log; timeout -k 10 500 bash tools/profile_gpu.sh $TAG > gpurun_out/prof_$TAG.log 2>&1; tail -1 gpurun_out/prof_$TAG.log; python tools/summarize_profile.py $TAG > /dev/null && timeout -k 10 300 python bench.py --no-build > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 200 gpurun_out/${TAG}_bench.json"
python tools/summarize_profile.py $TAG > /dev/null
cp gpurun_out/${TAG}_gputest.log profiles/${TAG}_gputest.log
tail -1 gpurun_out/${TAG}_bench.json > profiles/${TAG}_bench.json
python - <<PY
import json
d = json.load(open("profiles/${TAG}_bench.json"))
t = d["train_step"]
print("forward %.3f ms  %.0f traj/s  frac %.3f  traffic %s" % (d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"]))
print("train step %.2f ms  adjoint %.2f ms  forward+tape %.2f ms" % (t["ms_per_step"], t["roofline"]["adjoint"]["kernel_ms"], t["roofline"]["forward_with_tape"]["kernel_ms"]))
PY
python tools/fill_design.py $TAG
python tools/fill_readme.py $TAG > /dev/null
