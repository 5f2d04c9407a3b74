set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 ) > gpurun_out/r2a_tests.log 2>&1
( timeout 120 tools/ubench/pipes ) > gpurun_out/r2a_pipes.txt 2>&1
( for a in "0 1" "0 4" "0 8" "1 4" "1 8"; do timeout 200 python tools/bench_conv.py $a; done ) > gpurun_out/r2a_conv.txt 2>&1
( timeout 200 python tools/bench_attention.py 120000 2 bf16; timeout 200 python tools/bench_attention.py 960000 2 bf16; timeout 200 python tools/bench_attention.py 446000 4 bf16 ) > gpurun_out/r2a_attn.txt 2>&1
( timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline ) > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
( CDSEG_CONV_RG=0 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline ) > gpurun_out/r2a_bench_oldconv.json 2>> gpurun_out/r2a_bench.err
( timeout 400 python tools/cpu_sweep.py 24000 8 16 32 64 128 ) > gpurun_out/r2a_cpu_sweep.txt 2>&1
tail -5 gpurun_out/r2a_tests.log
cat gpurun_out/r2a_conv.txt
