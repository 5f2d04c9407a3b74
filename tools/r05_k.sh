# round 5, GPU call K: in-kernel occupancy stamps of the attention kernel with and without the tail zones (3 repeats each)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r05k_attn_timing.txt
for rep in 1 2 3; do
for kn in "0 0 0" "0 64 64"; do
  set -- $kn
  echo "== rep $rep LEAD=$1 TAIL1=$2 TAIL2=$3" >> gpurun_out/r05k_attn_timing.txt
  ( CDSEG_ATTN_LEAD=$1 CDSEG_ATTN_TAIL1=$2 CDSEG_ATTN_TAIL2=$3 CDSEG_ATTN_FORM=0 timeout 200 python tools/attn_timing.py 960000 2 2 ) 2>&1 | grep -v amdgpu >> gpurun_out/r05k_attn_timing.txt
done
done
grep "^==\|^launch\|waves resident\|wave life" gpurun_out/r05k_attn_timing.txt
