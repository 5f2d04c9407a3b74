# GPU: the native-Block-vs-oracle op test (both builds) and the IEEE-half error budget (profiles/r04_precision.txt)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -s -k "native_block_executor_vs_oracle" ) > gpurun_out/misc_tests.log 2>&1; tail -3 gpurun_out/misc_tests.log; grep "\[measure\]" gpurun_out/misc_tests.log | sed 's/^\.*//'
( timeout 600 python tools/bf16_budget.py 103000 fp16 ) > gpurun_out/r04_precision.txt 2> gpurun_out/r04_precision.err; tail -32 gpurun_out/r04_precision.txt; tail -3 gpurun_out/r04_precision.err
