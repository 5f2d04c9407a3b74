set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r2r_conv.txt
for sc in 8 4 1; do
  timeout 120 python tools/bench_conv.py 1 $sc 2>&1 | grep "conv level 1: weight" | sed "s/^/mappf ${sc}sc /" | cut -c1-220 >> gpurun_out/r2r_conv.txt
  timeout 120 python tools/bench_conv.py 0 $sc 2>&1 | grep "conv level 0: weight" | sed "s/^/mappf ${sc}sc /" | cut -c1-220 >> gpurun_out/r2r_conv.txt
done
cat gpurun_out/r2r_conv.txt
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv" ) > gpurun_out/r2r_tests.log 2>&1; tail -3 gpurun_out/r2r_tests.log
( timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-agreement ) > gpurun_out/r2r_bench.json 2> gpurun_out/r2r_bench.err
cut -c1-200 gpurun_out/r2r_bench.json
