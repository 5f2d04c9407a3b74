# Round-3 check of the IEEE-half build of the library: its op tests, the end-to-end tests, and the bench line next to the
# bfloat16 build's on the same box.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -s -k "float16 or f16" ) > gpurun_out/r3y_f16_ops.log 2>&1
tail -15 gpurun_out/r3y_f16_ops.log
( timeout 900 python -m pytest tests/test_gpu_e2e.py -q -s -k "fp16 or mixed_precision or full_size_properties_120k" ) > gpurun_out/r3y_f16_e2e.log 2>&1
tail -15 gpurun_out/r3y_f16_e2e.log
grep "\[measure\]" gpurun_out/r3y_f16_e2e.log | grep -i "fp16\|half" | sed 's/^\.*//'
( timeout 400 python bench.py --precision fp16+head --no-cpu-baseline --steps 10 ) > gpurun_out/r3y_bench_f16.json 2> gpurun_out/r3y_bench_f16.err
cat gpurun_out/r3y_bench_f16.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_forward'], d.get('agreement_vs_fp32'), d['single_scene_latency_ms'])"
tail -3 gpurun_out/r3y_bench_f16.err
( timeout 400 python bench.py --precision bf16+head --no-cpu-baseline --steps 10 ) > gpurun_out/r3y_bench_bf16.json 2> gpurun_out/r3y_bench_bf16.err
cat gpurun_out/r3y_bench_bf16.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_forward'], d.get('agreement_vs_fp32'), d['single_scene_latency_ms'])"
