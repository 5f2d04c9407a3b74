# Round-3: in-kernel phase timing + PMC passes of the attention kernel (new build and the round-2 build).
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 200 python tools/attn_timing.py 960000 2 2 ) > gpurun_out/r3b_attn_timing.txt 2>&1
( timeout 200 python tools/attn_timing.py 446000 4 2 ) >> gpurun_out/r3b_attn_timing.txt 2>&1
cat gpurun_out/r3b_attn_timing.txt
bash tools/pmc_r02.sh r3b_attn_new attn_bf16 python tools/bench_attention.py 960000 2 bf16 10 > /dev/null 2>&1
CDSEG_AB_LIB=tools/_ab/libcdseg_hip_r02attn.so bash tools/pmc_r02.sh r3b_attn_old attn_bf16 python tools/bench_attention.py 960000 2 bf16 10 > /dev/null 2>&1
cat gpurun_out/pmc_r3b_attn_new.txt gpurun_out/pmc_r3b_attn_old.txt
