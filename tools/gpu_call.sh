set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "block_rr or fused or native" 2>&1 | tail -12 ) > gpurun_out/r2h_tests.log 2>&1
( timeout 300 python tools/bench_block.py 8 ) > gpurun_out/r2h_block.txt 2>&1
( CDSEG_GEMM_DMA_BM=128 timeout 300 python tools/bench_gemm.py --scenes 8 | grep -E "n=6224|n=26912|sum" ) > gpurun_out/r2h_gemm128.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r2h_e2e.log 2>&1
( timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline ) > gpurun_out/r2h_bench.json 2> gpurun_out/r2h_bench.err
cat gpurun_out/r2h_tests.log gpurun_out/r2h_block.txt gpurun_out/r2h_gemm128.txt gpurun_out/r2h_e2e.log; tail -3 gpurun_out/r2h_bench.err; cut -c1-260 gpurun_out/r2h_bench.json
