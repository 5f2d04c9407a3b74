# round 6, GPU call AH: the kernel traces again (the validation run's traces included the round-5-setting leg)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_final -o run -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-agreement --no-kernel-timer --no-paper-pass > gpurun_out/r06i_bench_under_rocprof.json 2> gpurun_out/r06i_prof.err )
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_final_l1 -o run -- python bench.py --scenes-per-forward 8 --lanes 1 --serial --steps 6 --warmup 2 --no-cpu-baseline --no-agreement --no-kernel-timer --no-paper-pass > gpurun_out/r06i_bench_lanes1.json 2>> gpurun_out/r06i_prof.err )
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_final_l1b -o run -- python bench.py --lanes 1 --serial --steps 4 --warmup 2 --no-cpu-baseline --no-agreement --no-kernel-timer --no-paper-pass > gpurun_out/r06i_bench_lanes1_24.json 2>> gpurun_out/r06i_prof.err )
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_final -name "*.db" | head -1); python tools/prof_summary.py $DB 6 > gpurun_out/r06i_kernel_stats.txt 2>&1
DB=$(find /tmp/prof_final_l1 -name "*.db" | head -1); python tools/prof_summary.py $DB 8 > gpurun_out/r06i_lanes1_kernel_stats.txt 2>&1
DB=$(find /tmp/prof_final_l1b -name "*.db" | head -1); python tools/prof_summary.py $DB 6 > gpurun_out/r06i_lanes1_24_kernel_stats.txt 2>&1
head -24 gpurun_out/r06i_lanes1_kernel_stats.txt | cut -c1-150
head -24 gpurun_out/r06i_lanes1_24_kernel_stats.txt | cut -c1-150
tail -1 gpurun_out/r06i_bench_under_rocprof.json | cut -c1-200
echo "done at $SECONDS s"
