# same-box A B A B of the one-launch pooling (csrc/pool.hip) against the two-launch form, same library: bench.py with
# ops.pool_fused_ok patched to False for the "off" runs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for rep in 1 2; do
  for v in on off; do
    timeout 300 python -c "
import sys, runpy
import cdsegnet_amd.ops as o
if '$v' == 'off':
    o.pool_fused_ok = lambda *a: False
sys.argv = ['bench.py', '--steps', '10', '--warmup', '3', '--no-cpu-baseline', '--no-agreement']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value']/1e6,2), 'M points/s', round(d['ms_per_step'],2), 'ms/step; forward alone', round(d['roofline_forward']['wall_ms'],2), 'ms; bs=1', round(d['single_scene_latency_ms'],2), 'ms')"
  done
done
