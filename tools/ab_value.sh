#!/bin/bash
# same-box A/B of prebuilt library pairs (tools/_ab/<name>/), timed region only: bash tools/ab_value.sh <reps> <name> <name> ... [-- bench args]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
reps=$1; shift
names=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done
[ "$1" == "--" ] && shift
cp cdsegnet_amd/libcdseg_hip.so /tmp/keep_lib.so; cp cdsegnet_amd/libcdseg_hip_f16.so /tmp/keep_lib_f16.so
for rep in $(seq $reps); do
  for v in "${names[@]}"; do
    cp tools/_ab/$v/libcdseg_hip.so tools/_ab/$v/libcdseg_hip_f16.so cdsegnet_amd/
    touch cdsegnet_amd/libcdseg_hip.so cdsegnet_amd/libcdseg_hip_f16.so
    timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-agreement --no-kernel-timer "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value']/1e6,2), 'M points/s', round(d['ms_per_step'],2), 'ms/step')"
  done
done
cp /tmp/keep_lib.so cdsegnet_amd/libcdseg_hip.so; cp /tmp/keep_lib_f16.so cdsegnet_amd/libcdseg_hip_f16.so
