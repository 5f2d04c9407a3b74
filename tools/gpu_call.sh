set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "stem or sparse_conv or subm_conv3" 2>&1 | tail -8 ) > gpurun_out/r2f_tests.log 2>&1
( export CDSEG_BENCH_OLD_ONLY=1; for a in "2 8" "3 8" "4 8"; do timeout 200 python tools/bench_conv.py $a | grep "conv level"; CDSEG_CONV_DEEP_BM=0 timeout 200 python tools/bench_conv.py $a | grep "conv level"; done ) > gpurun_out/r2f_deepconv.txt 2>&1
( timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r2f_e2e.log 2>&1
( timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline ) > gpurun_out/r2f_bench.json 2> gpurun_out/r2f_bench.err
( CDSEG_STEM5=0 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-agreement ) > gpurun_out/r2f_bench_nostem5.json 2>> gpurun_out/r2f_bench.err
cat gpurun_out/r2f_tests.log gpurun_out/r2f_deepconv.txt gpurun_out/r2f_e2e.log; tail -3 gpurun_out/r2f_bench.err
