set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm or conv or stem or mlp or head or tail" 2>&1 | tail -12 ) > gpurun_out/r2g_tests.log 2>&1
( timeout 300 python tools/bench_gemm.py --scenes 8; CDSEG_GEMM_DMA=0 timeout 300 python tools/bench_gemm.py --scenes 8 ) > gpurun_out/r2g_gemm.txt 2>&1
( export CDSEG_BENCH_OLD_ONLY=1; for a in "2 8" "3 8" "4 8"; do for bm in 0 128 256; do CDSEG_GEMM_DMA_BM=$bm timeout 200 python tools/bench_conv.py $a | grep "conv level" | sed "s/^/DMA_BM=$bm /"; done; CDSEG_GEMM_DMA=0 CDSEG_CONV_DEEP_BM=0 timeout 200 python tools/bench_conv.py $a | grep "conv level" | sed "s/^/old /"; done ) > gpurun_out/r2g_deepconv.txt 2>&1
( timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-agreement ) > gpurun_out/r2g_bench.json 2> gpurun_out/r2g_bench.err
cat gpurun_out/r2g_tests.log gpurun_out/r2g_gemm.txt gpurun_out/r2g_deepconv.txt; tail -3 gpurun_out/r2g_bench.err; cat gpurun_out/r2g_bench.json | cut -c1-300
