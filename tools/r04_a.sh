#!/bin/bash
# round 4, first GPU call: deep-stage fused head / tail - parity tests, micro-benchmark, end-to-end bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "deep" 2>&1 | tail -30 > gpurun_out/r04a_tests.log
timeout 300 python tools/bench_deep.py 8 f16 > gpurun_out/r04a_bench_deep.txt 2>&1
timeout 300 python tools/bench_deep.py 1 f16 >> gpurun_out/r04a_bench_deep.txt 2>&1
timeout 300 python tools/bench_deep.py 8 bf16 >> gpurun_out/r04a_bench_deep.txt 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err
CDSEG_DUMMY=1 timeout 600 python bench.py --steps 8 --warmup 2 --lanes 1 --serial --no-cpu-baseline --no-agreement > gpurun_out/r04a_bench_lanes1.json 2>> gpurun_out/r04a_bench.err
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q 2>&1 | tail -15 > gpurun_out/r04a_e2e.log
tail -5 gpurun_out/r04a_tests.log; cat gpurun_out/r04a_bench_deep.txt; cat gpurun_out/r04a_bench.json | head -c 1500; echo; tail -3 gpurun_out/r04a_e2e.log
