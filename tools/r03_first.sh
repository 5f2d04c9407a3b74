# A/B: first query tile assigned statically and fetched next to the K/V DMAs; raised priority outside the key loops.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention" ) > gpurun_out/r3z_attn_tests.log 2>&1
tail -3 gpurun_out/r3z_attn_tests.log
{
for rep in 1 2; do
  for args in "960000 2 bf16 20 2 1" "864000 2 bf16 20 2 8" "432000 4 bf16 20 2 8" "112000 8 bf16 20 2 8" "27000 16 bf16 20 2 8" "103000 2 bf16 20 2 1" "51000 4 bf16 20 2 1"; do
    for lib in "" tools/_ab/libcdseg_hip_nostatic.so tools/_ab/libcdseg_hip_noprio.so tools/_ab/libcdseg_hip_prev.so; do
      CDSEG_AB_LIB=$lib timeout 60 python tools/bench_attention.py $args
    done
  done
done
} > gpurun_out/r3z_attn_first.txt 2>&1
grep "^attention\|Error\|error" gpurun_out/r3z_attn_first.txt | cut -c1-150
( CDSEG_ATTN_FORM=0 timeout 100 python tools/attn_timing.py 960000 2 2 ) > gpurun_out/r3z_attn_timing2.txt 2>&1
cat gpurun_out/r3z_attn_timing2.txt
