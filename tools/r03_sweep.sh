cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r3o_sweep.txt
for cfg in "8 3" "12 2" "24 1" "16 2" "8 4" "6 4" "12 3" "16 3"; do
  set -- $cfg
  echo "== scenes-per-forward $1 lanes $2" >> gpurun_out/r3o_sweep.txt
  timeout 300 python bench.py --scenes-per-forward $1 --lanes $2 --steps 10 --warmup 2 --no-cpu-baseline --no-agreement --no-kernel-timer 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value']/1e6, 'M points/s', r['ms_per_step'], 'ms/step', r['config']['scenes_per_step_per_gpu'], 'scenes/step')" >> gpurun_out/r3o_sweep.txt
done
cat gpurun_out/r3o_sweep.txt
