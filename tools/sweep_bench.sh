#!/bin/bash
# same-box sweep of bench.py configurations (24 scenes per step in every case): lanes x scenes per forward
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
for cfg in "8 3" "12 2" "6 4" "4 6" "24 1" "8 3"; do
  set -- $cfg
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-agreement --no-kernel-timer --scenes-per-forward $1 --lanes $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('scenes/forward $1 x lanes $2:', round(d['value']/1e6,2), 'M points/s', round(d['ms_per_step'],2), 'ms/step')"
done
