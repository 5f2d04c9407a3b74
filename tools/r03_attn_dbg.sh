set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
EXP=tools/_ab/libcdseg_hip_exp.so
{
for cfg in "1 0 1" "1 32 1" "0 0 1" "0 8 1" "0 0 2" "0 8 2" "0 8 4"; do
  set -- $cfg
  echo "== form $1 dbg $2 qsplit $3"
  for rep in 1 2; do
  CDSEG_ATTN_QSPLIT=$3 CDSEG_ATTN_DBG=$2 CDSEG_ATTN_FORM=$1 CDSEG_AB_LIB=$EXP timeout 120 python tools/bench_attention.py 960000 2 bf16 20 2
  CDSEG_ATTN_QSPLIT=$3 CDSEG_ATTN_DBG=$2 CDSEG_ATTN_FORM=$1 CDSEG_AB_LIB=$EXP timeout 120 python tools/bench_attention.py 446000 4 bf16 20 2
  done
done
echo "== timing persistent dbg 32"
CDSEG_ATTN_DBG=32 CDSEG_ATTN_FORM=1 timeout 200 python tools/attn_timing.py 960000 2 2
echo "== timing block dbg 8"
CDSEG_ATTN_DBG=8 CDSEG_ATTN_FORM=0 timeout 200 python tools/attn_timing.py 960000 2 2
} > gpurun_out/r3f_attn_dbg.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/r3f_attn_dbg.txt
