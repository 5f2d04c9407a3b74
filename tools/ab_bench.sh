#!/bin/bash
# same-box A/B of two prebuilt library pairs (tools/_ab/<name>/libcdseg_hip*.so): A B A B, short bench lines
# usage (on the GPU box): bash tools/ab_bench.sh on off [bench.py args]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
A=$1; B=$2; shift 2
mkdir -p gpurun_out
for rep in 1 2; do
  for v in $A $B; do
    cp tools/_ab/$v/libcdseg_hip.so tools/_ab/$v/libcdseg_hip_f16.so cdsegnet_amd/
    touch cdsegnet_amd/libcdseg_hip.so cdsegnet_amd/libcdseg_hip_f16.so
    timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-agreement "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value']/1e6,2), 'M points/s', round(d['ms_per_step'],2), 'ms/step; forward alone', round(d['roofline_forward']['wall_ms'],2), 'ms; bs=1', round(d['single_scene_latency_ms'],2), 'ms; attn frac', round(d['roofline']['frac'],4))"
  done
done
cp tools/_ab/$A/libcdseg_hip.so tools/_ab/$A/libcdseg_hip_f16.so cdsegnet_amd/
