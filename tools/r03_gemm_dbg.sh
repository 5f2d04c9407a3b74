cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r3t_gemm_dbg.txt
for dbg in 0 8 16 4; do
  echo "== CDSEG_GEMM_DBG=$dbg (8 = no global store in the plain bf16 epilogue, 16 = stores folded onto 1024 rows, 4 = no epilogue)" >> gpurun_out/r3t_gemm_dbg.txt
  CDSEG_GEMM_DBG=$dbg CDSEG_AB_LIB=tools/_ab/libcdseg_hip_gexp.so timeout 200 python tools/bench_gemm.py --scenes 8 2>&1 | grep -E "qkv|fc1 |sum" >> gpurun_out/r3t_gemm_dbg.txt
done
cat gpurun_out/r3t_gemm_dbg.txt
