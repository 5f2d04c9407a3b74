set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -s -k "mixed_precision or forward_work or model_variants or full_width" ) > gpurun_out/r3n_tests.log 2>&1
tail -5 gpurun_out/r3n_tests.log; grep "\[measure\]" gpurun_out/r3n_tests.log | tail -12
( timeout 600 python bench.py --no-cpu-baseline ) > gpurun_out/r3n_bench.json 2> gpurun_out/r3n_bench.err; tail -3 gpurun_out/r3n_bench.err; cat gpurun_out/r3n_bench.json
( timeout 300 python bench.py --shard 16 --dataset nuscenes --points 40000 --steps 5 --warmup 2 ) > gpurun_out/r3n_shard.json 2> gpurun_out/r3n_shard.err; tail -3 gpurun_out/r3n_shard.err; cat gpurun_out/r3n_shard.json
( timeout 600 python bench.py --protocol paper ) > gpurun_out/r3n_paper.json 2> gpurun_out/r3n_paper.err; tail -3 gpurun_out/r3n_paper.err; cat gpurun_out/r3n_paper.json
