set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
EXP=tools/_ab/libcdseg_hip_exp.so
( CDSEG_ATTN_FORM=0 CDSEG_AB_LIB=$EXP timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention" ) > gpurun_out/r3k_attn_tests.log 2>&1
tail -3 gpurun_out/r3k_attn_tests.log
{
for rep in 1 2; do
  for args in "960000 2 bf16 20 2" "446000 4 bf16 20 2" "114000 8 bf16 20 2" "27000 16 bf16 20 2" "778 32 bf16 20 2"; do
    for cfg in "0 1" "8 1" "8 2"; do
      set -- $cfg
      echo "== block dbg $1 qsplit $2"
      CDSEG_ATTN_QSPLIT=$2 CDSEG_ATTN_DBG=$1 CDSEG_ATTN_FORM=0 CDSEG_AB_LIB=$EXP timeout 60 python tools/bench_attention.py $args
    done
    echo "== flow"
    CDSEG_AB_LIB=$EXP timeout 60 python tools/bench_attention.py $args
  done
done
} > gpurun_out/r3k_attn_ab.txt 2>&1
grep "^attention\|Error\|error\|==" gpurun_out/r3k_attn_ab.txt
( CDSEG_ATTN_DBG=8 CDSEG_ATTN_FORM=0 timeout 100 python tools/attn_timing.py 960000 2 2 ) > gpurun_out/r3k_attn_timing.txt 2>&1
cat gpurun_out/r3k_attn_timing.txt
