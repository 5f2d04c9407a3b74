# round 6, GPU call X: fork stage A/B on the timed region
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06x
for rep in 1 2 3; do
  for f in 1 2; do
    echo -n "fork_stage=$f: "
    CDSEG_FORK_STAGE=$f timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-agreement --no-kernel-timer --no-paper-pass 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']/1e6,2), 'M points/s', round(d['ms_per_step'],2), 'ms/step')"
  done
done | tee ${O}_fork_ab.txt
echo "done at $SECONDS s"
