# A/B of a kernel change at op level (tools/bench_block.py with either library) and end to end
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "block_rr or native_block or deep_head" ) > gpurun_out/pf_tests.log 2>&1; tail -2 gpurun_out/pf_tests.log
for v in base pf base pf; do echo "--- $v"; CDSEG_AB_LIB=tools/_ab/$v/libcdseg_hip.so timeout 200 python tools/bench_block.py 8 2>&1 | grep -v amdgpu | head -6; done > gpurun_out/pf_block.txt 2>&1
cat gpurun_out/pf_block.txt
bash tools/ab_bench.sh pf base > gpurun_out/pf_ab.txt 2>&1; cat gpurun_out/pf_ab.txt
