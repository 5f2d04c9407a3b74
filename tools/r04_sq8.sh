# GPU run of the 8-wave 256 x 256 conv tile: parity tests, knob sweep on the conv shapes, K-loop stamps, e2e A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "deep_conv or gemm" ) > gpurun_out/sq8_tests.log 2>&1; tail -3 gpurun_out/sq8_tests.log
export CDSEG_AB_LIB=tools/_ab/libcdseg_hip_gexp.so
{
echo "--- level 2 (C = 128: 128 x 128 tiles)"; CDSEG_BENCH_OLD_ONLY=1 timeout 120 python tools/bench_conv.py 2 8 30 2>&1 | tail -1
for lvl in 3 4; do
  echo "--- level $lvl, 128 x 128 tiles (CDSEG_CONV_SQ=0)"; CDSEG_BENCH_OLD_ONLY=1 CDSEG_CONV_SQ=0 timeout 120 python tools/bench_conv.py $lvl 8 30 2>&1 | tail -1
  for alt in 0 1; do
    echo "--- level $lvl, 256 x 256 tiles of 8 waves, CDSEG_CONV_SQ_ALT=$alt"; CDSEG_BENCH_OLD_ONLY=1 CDSEG_CONV_SQ_ALT=$alt timeout 120 python tools/bench_conv.py $lvl 8 30 2>&1 | tail -1
  done
done
echo "--- single scenes (128 x 128 tiles)"
for lvl in 2 3 4; do CDSEG_BENCH_OLD_ONLY=1 timeout 120 python tools/bench_conv.py $lvl 1 30 2>&1 | tail -1; done
} > gpurun_out/sq8_bench_conv.txt 2>&1
cat gpurun_out/sq8_bench_conv.txt
unset CDSEG_AB_LIB
{
for alt in 1 0; do
  for lvl in 3 4; do echo "--- CDSEG_CONV_SQ_ALT=$alt"; CDSEG_CONV_SQ_ALT=$alt timeout 120 python tools/conv_timing.py $lvl 8 2>&1 | tail -2; done
done
echo "--- level 2"; timeout 120 python tools/conv_timing.py 2 8 2>&1 | tail -2
} > gpurun_out/sq8_conv_timing.txt 2>&1
cat gpurun_out/sq8_conv_timing.txt
bash tools/ab_bench.sh mask sq8 > gpurun_out/sq8_ab.txt 2>&1; cat gpurun_out/sq8_ab.txt
