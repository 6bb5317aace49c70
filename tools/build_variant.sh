#!/bin/bash
# An experiment build of the product library next to it: tools/build_variant.sh NAME -DFLAG ...  ->  hode/lab/libhode_NAME.so
# (git-ignored; load with HODE_LIB=<path>; the A/B timings of DESIGN.md section 6.2 come from pairs built this way, timed on ONE box).
set -e
name=$1; shift
cd "$(dirname "$0")/../hybrid-ode-for-glp-1-and-glucose_amd/csrc"
obj=/tmp/hode_variant_$name; mkdir -p $obj ../hode/lab
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function "$@" -c $f -o $obj/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../hode/lab/libhode_$name.so $obj/*.o
echo "built hode/lab/libhode_$name.so ($*)"
