# round 5, GPU call H: C = 128 sparse conv on the 8-wave 256 x 128 tile (CDSEG_CONV_SQ128_MIN_M) vs the 128 x 128 tile
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=tools/_ab/libcdseg_hip_expgemm.so
: > gpurun_out/r05h_conv128.txt
for rep in 1 2; do
for m in 0 20000; do
  echo "== 8 scenes, CDSEG_CONV_SQ128_MIN_M=$m" >> gpurun_out/r05h_conv128.txt
  ( CDSEG_AB_LIB=$L CDSEG_CONV_SQ128_MIN_M=$m timeout 200 python tools/bench_conv.py 2 8 30 ) >> gpurun_out/r05h_conv128.txt 2>&1
done
done
for m in 0 8000; do
  echo "== 1 scene, CDSEG_CONV_SQ128_MIN_M=$m" >> gpurun_out/r05h_conv128.txt
  ( CDSEG_AB_LIB=$L CDSEG_CONV_SQ128_MIN_M=$m timeout 200 python tools/bench_conv.py 2 1 30 ) >> gpurun_out/r05h_conv128.txt 2>&1
done
for m in 0 20000; do
  echo "== 24 scenes, CDSEG_CONV_SQ128_MIN_M=$m" >> gpurun_out/r05h_conv128.txt
  ( CDSEG_AB_LIB=$L CDSEG_CONV_SQ128_MIN_M=$m timeout 200 python tools/bench_conv.py 2 24 20 ) >> gpurun_out/r05h_conv128.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/r05h_conv128.txt
