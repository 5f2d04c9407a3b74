set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention or subm_conv3 or sparse_conv" 2>&1 | tail -15 ) > gpurun_out/r2b_tests.log 2>&1
( timeout 200 python tools/bench_attention.py 120000 2 bf16; timeout 200 python tools/bench_attention.py 960000 2 bf16; timeout 200 python tools/bench_attention.py 446000 4 bf16 ) > gpurun_out/r2b_attn.txt 2>&1
( for d in 0 1 2 3; do echo "DBG=$d"; CDSEG_CONV_DBG=$d CDSEG_BENCH_NEW_ONLY=1 timeout 200 python tools/bench_conv.py 1 8 | grep weight; CDSEG_CONV_DBG=$d CDSEG_BENCH_NEW_ONLY=1 timeout 200 python tools/bench_conv.py 0 8 | grep weight; done ) > gpurun_out/r2b_conv_dbg.txt 2>&1
export CDSEG_BENCH_NEW_ONLY=1
bash tools/pmc_r02.sh conv32 conv_rg_kernel python tools/bench_conv.py 0 8 10 > /dev/null 2>&1
bash tools/pmc_r02.sh conv64 conv_rg_kernel python tools/bench_conv.py 1 8 10 > /dev/null 2>&1
bash tools/pmc_r02.sh attn attn_bf16 python tools/bench_attention.py 960000 2 bf16 10 > /dev/null 2>&1
unset CDSEG_BENCH_NEW_ONLY
( rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TA|TCC|SQ|TD)_[A-Z0-9_a-z]+" | sort -u | tr '\n' ' ' ) > gpurun_out/r2b_counters.txt 2>&1
( timeout 600 python bench.py --steps 8 --warmup 3 --cpu-points 120000 --cpu-threads 16 ) > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err
cat gpurun_out/r2b_tests.log gpurun_out/r2b_attn.txt gpurun_out/r2b_conv_dbg.txt
tail -3 gpurun_out/r2b_bench.err
