# round 5, GPU call D: full GPU suite, smoke, default bench line, host contention (8 issuers on one GPU), PMC traffic
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q -s ) > gpurun_out/r05d_tests.log 2>&1
tail -4 gpurun_out/r05d_tests.log
grep "\[measure\]" gpurun_out/r05d_tests.log | sed 's/^\.*//' > gpurun_out/r05d_parity_measured.txt; wc -l gpurun_out/r05d_parity_measured.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r05d_smoke.log 2>&1; tail -5 gpurun_out/r05d_smoke.log
( timeout 900 python bench.py ) > gpurun_out/r05d_bench.json 2> gpurun_out/r05d_bench.err
head -c 300 gpurun_out/r05d_bench.json; echo; tail -3 gpurun_out/r05d_bench.err
( timeout 600 python tools/host_contention.py 8 40000 8 20 ) > gpurun_out/r05d_host_contention.txt 2>&1
grep "^ranks\|^#" gpurun_out/r05d_host_contention.txt
bash tools/pmc_bench_traffic.sh $GRAFT_REPO_ROOT/gpurun_out/r05d_attention_traffic.json > gpurun_out/r05d_pmc.log 2>&1
tail -14 gpurun_out/r05d_pmc.log
