set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -s -k "degenerate" ) > gpurun_out/r2w_tests.log 2>&1; tail -15 gpurun_out/r2w_tests.log | cut -c1-220
