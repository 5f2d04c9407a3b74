set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention" ) > gpurun_out/r3q_attn_tests.log 2>&1
tail -3 gpurun_out/r3q_attn_tests.log
{
for rep in 1 2 3; do
  for args in "960000 2 bf16 20 2" "446000 4 bf16 20 2" "114000 8 bf16 20 2" "27000 16 bf16 20 2" "6200 32 bf16 20 2"; do
    timeout 60 python tools/bench_attention.py $args
    CDSEG_AB_LIB=tools/_ab/libcdseg_hip_two.so timeout 60 python tools/bench_attention.py $args
  done
done
} > gpurun_out/r3q_attn_ab.txt 2>&1
grep "^attention\|Error\|error" gpurun_out/r3q_attn_ab.txt
( CDSEG_ATTN_FORM=0 timeout 100 python tools/attn_timing.py 960000 2 2 ) > gpurun_out/r3q_attn_timing.txt 2>&1
cat gpurun_out/r3q_attn_timing.txt
