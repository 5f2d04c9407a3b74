# round 5, GPU call L: cpe_head_fused with weights / residual rows requested a phase ahead - tests and A/B (bfloat16 build)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "cpe_head or block_executor" ) 2>&1 | tail -2
: > gpurun_out/r05l_head.txt
for rep in 1 2 3; do
  for lib in "" tools/_ab/libcdseg_hip_nopf.so; do
    echo "== lib=${lib:-product(prefetch)}" >> gpurun_out/r05l_head.txt
    ( CDSEG_AB_LIB=$lib timeout 200 python tools/bench_block.py 8 ) 2>&1 | grep "^head" >> gpurun_out/r05l_head.txt
  done
done
cat gpurun_out/r05l_head.txt
