cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 tools/ubench/attn_loop > gpurun_out/r3g_attn_loop.txt 2>&1
cat gpurun_out/r3g_attn_loop.txt
