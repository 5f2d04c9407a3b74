# Round-end validation on the GPU box: full GPU test suite (with the [measure] lines), smoke, PMC traffic of bench.py's own
# forwards, the default bench line, kernel-trace profiles of the bench and of single-scene inference.
# usage: gpurun --timeout 3000 -- bash tools/gpu_call.sh   (outputs under gpurun_out/, copied to profiles/ by hand)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q -s ) > gpurun_out/final_tests.log 2>&1
tail -3 gpurun_out/final_tests.log
grep "\[measure\]" gpurun_out/final_tests.log | sed 's/^\.*//' > gpurun_out/r02_parity_measured.txt; wc -l gpurun_out/r02_parity_measured.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/final_smoke.log 2>&1; tail -2 gpurun_out/final_smoke.log
bash tools/pmc_bench_traffic.sh $GRAFT_REPO_ROOT/gpurun_out/r02_attention_traffic.json > gpurun_out/final_pmc.log 2>&1
tail -12 gpurun_out/final_pmc.log
cp gpurun_out/r02_attention_traffic.json profiles/r02_attention_traffic.json
( timeout 600 python bench.py ) > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
cat gpurun_out/final_bench.json
cd /tmp && export TMPDIR=/tmp
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_final -o run -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-agreement --no-kernel-timer > gpurun_out/final_bench_under_rocprof.json 2> gpurun_out/final_prof.err )
( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_final1 -o run -- python tools/single_scene_profile.py > gpurun_out/final_single.txt 2> gpurun_out/final_single.err )
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_final -name "*.db" | head -1)
python tools/prof_summary.py $DB 6 > gpurun_out/final_kernel_stats.txt 2>&1
DB=$(find /tmp/prof_final1 -name "*.db" | head -1)
python tools/prof_summary.py $DB 30 > gpurun_out/final_single_kernel_stats.txt 2>&1
cat gpurun_out/final_single.txt; head -30 gpurun_out/final_kernel_stats.txt | cut -c1-150
