#!/bin/bash
# Build a named pair of library variants (bfloat16 + IEEE-half builds) for same-box A/B runs with tools/ab_bench.sh:
#   bash tools/build_variant.sh <name> [-DFLAG ...]   ->  tools/_ab/<name>/libcdseg_hip.so, libcdseg_hip_f16.so
# Every source is recompiled with the extra flags (a flag only matters in the file that reads it).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=tools/_ab/$name; mkdir -p $out/o $out/o16
SRCS=$(python -c "from cdsegnet_amd import build; print(' '.join(build.SOURCES))")
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC"
for s in $SRCS; do
  $H "$@" -c cdsegnet_amd/csrc/$s -o $out/o/${s%.hip}.o &
  $H "$@" -DCDSEG_LP_F16 -c cdsegnet_amd/csrc/$s -o $out/o16/${s%.hip}.o &
done
wait
$H -shared -o $out/libcdseg_hip.so $out/o/*.o
$H -shared -o $out/libcdseg_hip_f16.so $out/o16/*.o
rm -rf $out/o $out/o16
ls -la $out
