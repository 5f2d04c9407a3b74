set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-agreement ) > gpurun_out/r2i_bench.json 2> gpurun_out/r2i_bench.err
cd /tmp && export TMPDIR=/tmp
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_r2i -o run -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-agreement --no-kernel-timer > gpurun_out/r2i_bench_under_rocprof.json 2> gpurun_out/r2i_prof.err )
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_r2i -name "*.db" | head -1)
python tools/prof_summary.py $DB 6 > gpurun_out/r2i_kernel_stats.txt 2>&1
cut -c1-200 gpurun_out/r2i_bench.json; head -45 gpurun_out/r2i_kernel_stats.txt
