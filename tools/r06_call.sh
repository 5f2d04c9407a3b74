# round 6, GPU call K: x3 tests again on the final dispatch; kernel traces of the 16-bit default (bs = 1 and 8 scenes, one lane);
# bench line incl. the fp32x3 leg
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06k
( timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_ops.py -x -q -k "fp32x3" ) > ${O}_tests.log 2>&1; tail -2 ${O}_tests.log
cd /tmp && export TMPDIR=/tmp
( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s1 -o run -- python tools/single_scene_profile.py > /tmp/s1.txt 2> /tmp/s1.err )
( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_l1 -o run -- python bench.py --steps 5 --warmup 3 --lanes 1 --serial --no-cpu-baseline --no-agreement --no-paper-pass --no-kernel-timer > /tmp/l1.txt 2> /tmp/l1.err )
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_s1 -name "*.db" | head -1); python tools/prof_summary.py $DB 30 > ${O}_single_kernel_stats.txt 2>&1
DB=$(find /tmp/prof_l1 -name "*.db" | head -1); python tools/prof_summary.py $DB 8 > ${O}_lanes1_kernel_stats.txt 2>&1
grep -v amdgpu /tmp/s1.txt | tail -1; head -34 ${O}_single_kernel_stats.txt | cut -c1-150
head -34 ${O}_lanes1_kernel_stats.txt | cut -c1-150
( timeout 600 python bench.py ) > ${O}_bench.json 2> ${O}_bench.err
python3 -c "
import json
d=json.loads(open('${O}_bench.json').read().strip().splitlines()[-1])
print('e2e', round(d['value']/1e6,2), 'M; attn frac', round(d['roofline']['frac'],4), 'bs1', d['headline']['bs1_ms_per_scene'])
print(json.dumps(d['parity_mode'])[:900])"
echo "done at $SECONDS s"
