# round 5, GPU call B: attention schedule sweep (same process, three builds), in-kernel occupancy stamps for two settings
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AB=tools/_ab
( timeout 600 python tools/attn_sweep.py --libs base=$AB/libcdseg_hip_base.so,exp=$AB/libcdseg_hip_exp.so,ns16=$AB/libcdseg_hip_exp_ns16.so \
   --knobs "0,0,0;0,0,64;0,64,64;0,64,128;0,128,128;32,0,0;32,64,64;64,64,128;0,32,32;0,0,128;0,128,256;0,0,0,2;0,0,0,4" ) > gpurun_out/r05b_attn_sweep.txt 2>&1
grep -c median gpurun_out/r05b_attn_sweep.txt
for kn in "0 0 0" "0 64 64" "32 64 64" "0 128 128"; do
  set -- $kn
  echo "== timing LEAD=$1 TAIL1=$2 TAIL2=$3" >> gpurun_out/r05b_attn_timing.txt
  ( CDSEG_ATTN_LEAD=$1 CDSEG_ATTN_TAIL1=$2 CDSEG_ATTN_TAIL2=$3 CDSEG_ATTN_FORM=0 timeout 200 python tools/attn_timing.py 960000 2 2 ) >> gpurun_out/r05b_attn_timing.txt 2>&1
done
( timeout 300 python -m pytest tests/test_gpu_e2e.py -x -q -k "half_build or half_trunk" ) > gpurun_out/r05b_tests.log 2>&1
tail -3 gpurun_out/r05b_tests.log
