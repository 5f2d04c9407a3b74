# round 6, GPU call G: where does the fp32x3 mode spend its time (kernel trace)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06g
cd /tmp && export TMPDIR=/tmp
( cd $GRAFT_REPO_ROOT && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_x3 -o run -- python tools/parity_mode_profile.py fp32x3 8 > /tmp/x3.txt 2> /tmp/x3.err )
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_x3 -name "*.db" | head -1); python tools/prof_summary.py $DB 6 > ${O}_x3_kernel_stats.txt 2>&1
head -45 ${O}_x3_kernel_stats.txt; cat /tmp/x3.txt | grep -v amdgpu
( timeout 300 python tools/parity_mode_profile.py fp32x3 8; timeout 300 python tools/parity_mode_profile.py fp32 4 ) 2>&1 | grep -v amdgpu
echo "done at $SECONDS s"
