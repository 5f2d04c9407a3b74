set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r2x_stem.txt
for rb in 2 3; do
  CDSEG_STEM_RB=$rb timeout 200 python tools/bench_stem.py 8 2>&1 | grep "stem5 n=" | sed "s/^/rb=$rb /" >> gpurun_out/r2x_stem.txt
  CDSEG_STEM_RB=$rb timeout 200 python tools/bench_stem.py 1 2>&1 | grep "stem5 n=" | sed "s/^/rb=$rb /" >> gpurun_out/r2x_stem.txt
done
cat gpurun_out/r2x_stem.txt
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -s -k "stem5" ) > gpurun_out/r2x_tests.log 2>&1; tail -8 gpurun_out/r2x_tests.log | cut -c1-160
for rb in 2 3; do
( CDSEG_STEM_RB=$rb timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-agreement ) > gpurun_out/r2x_bench_rb$rb.json 2> gpurun_out/r2x_bench.err
cut -c1-200 gpurun_out/r2x_bench_rb$rb.json
done
