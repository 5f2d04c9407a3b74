set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python tools/bench_next_rows.py ) > gpurun_out/r02_next_rows.txt 2> gpurun_out/r2t_next.err
cat gpurun_out/r02_next_rows.txt; tail -3 gpurun_out/r2t_next.err
rm -f gpurun_out/r02_other_configs.txt
for cfg in "--dataset scannet200" "--dataset nuscenes" "--robust" "--precision fp32 --scenes-per-forward 4"; do
  ( timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-agreement $cfg ) 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bench.py $cfg :', round(d['value']/1e6,2), 'M points/s,', round(d['ms_per_step'],2), 'ms/step,', d['config']['scenes_per_step_per_gpu'], 'scenes/step, mean points/scene', round(d['config']['points_per_scene_mean']), ', attention frac', round(d.get('roofline',{}).get('frac',0),4), ', single-scene latency ms', round(d.get('single_scene_latency_ms',0),2))" >> gpurun_out/r02_other_configs.txt
done
cat gpurun_out/r02_other_configs.txt
