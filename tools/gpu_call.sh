set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r2c_tests.log 2>&1
( timeout 120 tools/ubench/pipes | grep -E "exp_f16|legacy|16 perm|bpermute|16 exp  |16 fma" ) > gpurun_out/r2c_pipes.txt 2>&1
cd /tmp && export TMPDIR=/tmp
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_r2c -o run -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-agreement --no-kernel-timer > gpurun_out/r2c_bench_under_rocprof.json 2> gpurun_out/r2c_prof.err )
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_r2c -name "*.db" | head -1)
python tools/prof_summary.py $DB 6 > gpurun_out/r2c_kernel_stats.txt 2>&1
tail -5 gpurun_out/r2c_tests.log; cat gpurun_out/r2c_pipes.txt; head -40 gpurun_out/r2c_kernel_stats.txt
