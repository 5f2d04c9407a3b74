set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "subm_conv3 or sparse_conv" 2>&1 | tail -15 ) > gpurun_out/r2d_tests.log 2>&1
( export CDSEG_BENCH_NEW_ONLY=1; for a in "0 1" "0 4" "0 8" "1 4" "1 8"; do timeout 200 python tools/bench_conv.py $a | grep weight; done ) > gpurun_out/r2d_conv.txt 2>&1
( timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-agreement ) > gpurun_out/r2d_bench.json 2> gpurun_out/r2d_bench.err
cat gpurun_out/r2d_tests.log gpurun_out/r2d_conv.txt; tail -2 gpurun_out/r2d_bench.err
