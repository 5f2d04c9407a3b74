# round 5, last GPU call: the full GPU suite, smoke and the default bench line on the final tree
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q -s ) > gpurun_out/r05m_tests.log 2>&1
tail -3 gpurun_out/r05m_tests.log
grep "\[measure\]" gpurun_out/r05m_tests.log | sed 's/^\.*//' > gpurun_out/r05m_parity_measured.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r05m_smoke.log 2>&1; tail -5 gpurun_out/r05m_smoke.log
( timeout 900 python bench.py ) > gpurun_out/r05m_bench.json 2> gpurun_out/r05m_bench.err
head -c 300 gpurun_out/r05m_bench.json; echo
