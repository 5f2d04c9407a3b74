set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r2j_tests.log 2>&1
tail -3 gpurun_out/r2j_tests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r2j_smoke.log 2>&1; tail -2 gpurun_out/r2j_smoke.log
bash tools/pmc_bench_traffic.sh $GRAFT_REPO_ROOT/gpurun_out/r02_attention_traffic.json > gpurun_out/r2j_pmc.log 2>&1
tail -14 gpurun_out/r2j_pmc.log
mkdir -p profiles; cp gpurun_out/r02_attention_traffic.json profiles/r02_attention_traffic.json
( timeout 600 python bench.py ) > gpurun_out/r2j_bench.json 2> gpurun_out/r2j_bench.err
cat gpurun_out/r2j_bench.json
cd /tmp && export TMPDIR=/tmp
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_r2j -o run -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-agreement --no-kernel-timer > gpurun_out/r2j_bench_under_rocprof.json 2> gpurun_out/r2j_prof.err )
( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_r2j1 -o run -- python tools/single_scene_profile.py > gpurun_out/r2j_single.txt 2> gpurun_out/r2j_single.err )
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_r2j -name "*.db" | head -1)
python tools/prof_summary.py $DB 6 > gpurun_out/r2j_kernel_stats.txt 2>&1
DB=$(find /tmp/prof_r2j1 -name "*.db" | head -1)
python tools/prof_summary.py $DB 30 > gpurun_out/r2j_single_kernel_stats.txt 2>&1
cat gpurun_out/r2j_single.txt; head -3 gpurun_out/r2j_single_kernel_stats.txt
