set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q -s ) > gpurun_out/r2u_tests.log 2>&1
tail -3 gpurun_out/r2u_tests.log
grep "\[measure\]" gpurun_out/r2u_tests.log > gpurun_out/r02_parity_measured.txt; wc -l gpurun_out/r02_parity_measured.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r2u_smoke.log 2>&1; tail -2 gpurun_out/r2u_smoke.log
bash tools/pmc_bench_traffic.sh $GRAFT_REPO_ROOT/gpurun_out/r02_attention_traffic.json > gpurun_out/r2u_pmc.log 2>&1
tail -12 gpurun_out/r2u_pmc.log
cp gpurun_out/r02_attention_traffic.json profiles/r02_attention_traffic.json
( timeout 600 python bench.py ) > gpurun_out/r2u_bench.json 2> gpurun_out/r2u_bench.err
cat gpurun_out/r2u_bench.json
cd /tmp && export TMPDIR=/tmp
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_r2u -o run -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-agreement --no-kernel-timer > gpurun_out/r2u_bench_under_rocprof.json 2> gpurun_out/r2u_prof.err )
( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_r2u1 -o run -- python tools/single_scene_profile.py > gpurun_out/r2u_single.txt 2> gpurun_out/r2u_single.err )
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_r2u -name "*.db" | head -1)
python tools/prof_summary.py $DB 6 > gpurun_out/r2u_kernel_stats.txt 2>&1
DB=$(find /tmp/prof_r2u1 -name "*.db" | head -1)
python tools/prof_summary.py $DB 30 > gpurun_out/r2u_single_kernel_stats.txt 2>&1
cat gpurun_out/r2u_single.txt; head -30 gpurun_out/r2u_kernel_stats.txt | cut -c1-150
CDSEG_BENCH_NEW_ONLY=1 bash tools/pmc_r02.sh conv32 conv_ll python tools/bench_conv.py 0 8 > /dev/null 2>&1
CDSEG_BENCH_NEW_ONLY=1 bash tools/pmc_r02.sh conv64 conv_ll python tools/bench_conv.py 1 8 > /dev/null 2>&1
head -40 gpurun_out/pmc_conv64.txt | cut -c1-120
