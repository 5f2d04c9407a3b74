cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06aa
( timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -k "native_plan or degenerate" ) > ${O}_tests_plan.log 2>&1; tail -15 ${O}_tests_plan.log
echo "done at $SECONDS s"
