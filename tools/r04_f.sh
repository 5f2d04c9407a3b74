#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/r04f_conv_bm256.txt
echo "--- CDSEG_GEMM_DMA_BM=256 (256-row tiles, 16 waves, one block per CU), timing build" >> gpurun_out/r04f_conv_bm256.txt
for lv in 2 3 4; do CDSEG_GEMM_DMA_BM=256 timeout 200 python tools/conv_timing.py $lv 8 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04f_conv_bm256.txt; done
for lv in 3 4; do CDSEG_GEMM_DMA_BM=256 timeout 200 python tools/conv_timing.py $lv 1 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04f_conv_bm256.txt; done
echo "--- default (128-row tiles), timing build" >> gpurun_out/r04f_conv_bm256.txt
for lv in 3 4; do timeout 200 python tools/conv_timing.py $lv 8 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04f_conv_bm256.txt; done
cat gpurun_out/r04f_conv_bm256.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r04f_tests.log; tail -3 gpurun_out/r04f_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04f_bench.json 2> gpurun_out/r04f_bench.err; head -c 300 gpurun_out/r04f_bench.json; echo
