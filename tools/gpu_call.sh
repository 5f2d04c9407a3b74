set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/bench_block.py 8 > gpurun_out/r2v_block.txt 2>&1
cat gpurun_out/r2v_block.txt
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "block_rr or conv3" ) > gpurun_out/r2v_tests.log 2>&1; tail -3 gpurun_out/r2v_tests.log
( timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-agreement ) > gpurun_out/r2v_bench.json 2> gpurun_out/r2v_bench.err
cut -c1-200 gpurun_out/r2v_bench.json
( CDSEG_BLOCK_RR_HEAD=1 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-agreement ) > gpurun_out/r2v_bench_head.json 2> gpurun_out/r2v_bench.err
cut -c1-200 gpurun_out/r2v_bench_head.json
