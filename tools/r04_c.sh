#!/bin/bash
# round 4, third GPU call: polynomial GELU everywhere a 16-bit result follows; conv group skipping with scalar masks (wide tiles off)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "deep or conv or gemm or mlp or tail or head or rr" 2>&1 | tail -40 > gpurun_out/r04c_tests.log
timeout 300 python tools/bench_deep.py 8 f16 > gpurun_out/r04c_bench_deep.txt 2>&1
timeout 300 python tools/bench_deep.py 1 f16 >> gpurun_out/r04c_bench_deep.txt 2>&1
timeout 300 python tools/bench_block.py 8 > gpurun_out/r04c_bench_block.txt 2>&1
for lv in 2 3 4; do for sc in 8 1; do timeout 300 python tools/bench_conv.py $lv $sc 20 2>&1 | grep "conv level" ; done; done > gpurun_out/r04c_bench_conv.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04c_bench.json 2> gpurun_out/r04c_bench.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_l1 -o l1 -- python $GRAFT_REPO_ROOT/bench.py --lanes 1 --serial --steps 6 --warmup 2 --no-cpu-baseline --no-agreement --no-kernel-timer > $GRAFT_REPO_ROOT/gpurun_out/r04c_bench_lanes1.json 2> $GRAFT_REPO_ROOT/gpurun_out/r04c_prof.err
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_l1 -name "*.db" | head -1); python tools/prof_summary.py $DB 8 > gpurun_out/r04c_lanes1_kernel_stats.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q 2>&1 | tail -15 > gpurun_out/r04c_e2e.log
tail -5 gpurun_out/r04c_tests.log; cat gpurun_out/r04c_bench_deep.txt gpurun_out/r04c_bench_conv.txt gpurun_out/r04c_bench_block.txt; head -c 300 gpurun_out/r04c_bench.json; echo; tail -3 gpurun_out/r04c_e2e.log; head -34 gpurun_out/r04c_lanes1_kernel_stats.txt | cut -c1-150
