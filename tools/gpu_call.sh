set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r2l_convll.txt
for ll in 0 1 2; do
  CDSEG_CONV_LL=$ll timeout 120 python tools/bench_conv.py 0 8 2>&1 | grep "conv level 0: weight" | sed "s/^/ll=$ll /" >> gpurun_out/r2l_convll.txt
  CDSEG_CONV_LL=$ll timeout 120 python tools/bench_conv.py 0 1 2>&1 | grep "conv level 0: weight" | sed "s/^/ll=$ll 1scene /" >> gpurun_out/r2l_convll.txt
done
for ll in 0 1 2 3; do
  CDSEG_CONV_LL=$ll timeout 120 python tools/bench_conv.py 1 8 2>&1 | grep "conv level 1: weight" | sed "s/^/ll=$ll /" >> gpurun_out/r2l_convll.txt
  CDSEG_CONV_LL=$ll timeout 120 python tools/bench_conv.py 1 1 2>&1 | grep "conv level 1: weight" | sed "s/^/ll=$ll 1scene /" >> gpurun_out/r2l_convll.txt
done
cat gpurun_out/r2l_convll.txt | cut -c1-200
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv" ) > gpurun_out/r2l_tests.log 2>&1; tail -3 gpurun_out/r2l_tests.log
for ll in 1 2; do
( CDSEG_CONV_LL=$ll timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-agreement ) > gpurun_out/r2l_bench_ll$ll.json 2> gpurun_out/r2l_bench.err
cut -c1-200 gpurun_out/r2l_bench_ll$ll.json
done
