# Round-3 checkpoint on the GPU box: full GPU test suite, smoke, default bench line.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r3}
( timeout 1500 python -m pytest tests -m gpu -x -q -s ) > gpurun_out/${TAG}_tests.log 2>&1
tail -3 gpurun_out/${TAG}_tests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log
( timeout 600 python bench.py ${BENCH_ARGS} ) > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cat gpurun_out/${TAG}_bench.json
