set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for d in 1 2 3 4; do
  CDSEG_CONV_DEPTH=$d CDSEG_BENCH_NEW_ONLY=1 timeout 120 python tools/bench_conv.py 0 8 2>&1 | grep "conv level" | sed "s/^/depth=$d /" >> gpurun_out/r2k_convdepth.txt
  CDSEG_CONV_DEPTH=$d CDSEG_BENCH_NEW_ONLY=1 timeout 120 python tools/bench_conv.py 1 8 2>&1 | grep "conv level" | sed "s/^/depth=$d /" >> gpurun_out/r2k_convdepth.txt
done
cat gpurun_out/r2k_convdepth.txt | cut -c1-200
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv" ) > gpurun_out/r2k_tests.log 2>&1; tail -3 gpurun_out/r2k_tests.log
bash tools/pmc_bench_traffic.sh $GRAFT_REPO_ROOT/gpurun_out/r02_attention_traffic.json > gpurun_out/r2k_pmc.log 2>&1
tail -14 gpurun_out/r2k_pmc.log
cp gpurun_out/r02_attention_traffic.json profiles/r02_attention_traffic.json
( timeout 600 python bench.py ) > gpurun_out/r2k_bench.json 2> gpurun_out/r2k_bench.err
cat gpurun_out/r2k_bench.json; tail -3 gpurun_out/r2k_bench.err
