#!/bin/bash
# round 4: where a K step of the gathered conv goes (in-kernel stamps, tools/conv_timing.py)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "conv or gemm" 2>&1 | tail -4 > gpurun_out/r04e_tests.log
: > gpurun_out/r04e_conv_timing.txt
for lv in 2 3 4; do timeout 200 python tools/conv_timing.py $lv 8 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04e_conv_timing.txt; done
echo "--- CDSEG_GEMM_ALT=0 (every wave issues before it multiplies)" >> gpurun_out/r04e_conv_timing.txt
for lv in 2 3 4; do CDSEG_GEMM_ALT=0 timeout 200 python tools/conv_timing.py $lv 8 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04e_conv_timing.txt; done
echo "--- CDSEG_CONV_WIDE=1 (256-column tiles, three stages, one block per CU)" >> gpurun_out/r04e_conv_timing.txt
for lv in 3 4; do CDSEG_CONV_WIDE=1 timeout 200 python tools/conv_timing.py $lv 8 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04e_conv_timing.txt; done
echo "--- single scene" >> gpurun_out/r04e_conv_timing.txt
for lv in 2 3 4; do timeout 200 python tools/conv_timing.py $lv 1 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04e_conv_timing.txt; done
echo "--- product build, tools/bench_conv.py" >> gpurun_out/r04e_conv_timing.txt
for lv in 2 3 4; do for sc in 8 1; do timeout 300 python tools/bench_conv.py $lv $sc 20 2>&1 | grep "conv level" ; done; done >> gpurun_out/r04e_conv_timing.txt
timeout 300 python tools/bench_gemm.py --scenes 8 2>&1 | grep -v amdgpu > gpurun_out/r04e_gemm_shapes.txt
tail -3 gpurun_out/r04e_tests.log; cat gpurun_out/r04e_conv_timing.txt; tail -12 gpurun_out/r04e_gemm_shapes.txt
