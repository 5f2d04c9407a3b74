cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm or conv or linear or mlp or block" ) > gpurun_out/r3s_gemm_tests.log 2>&1
tail -3 gpurun_out/r3s_gemm_tests.log
timeout 200 python tools/bench_gemm.py --scenes 8 > gpurun_out/r3s_gemm8.txt 2>&1; cat gpurun_out/r3s_gemm8.txt
for a in "2 8" "3 8" "4 8"; do timeout 100 python tools/bench_conv.py $a 2>&1 | tail -3; done > gpurun_out/r3s_conv.txt 2>&1; cat gpurun_out/r3s_conv.txt
