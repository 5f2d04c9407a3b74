# round 5, GPU call E: C = 512 fused head / tail (tests, micro-bench, launch census, bench), host contention with blocking
# host reads, attention HBM traffic by launch shape
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "deep_head or block_executor or attention" ) > gpurun_out/r05e_tests_ops.log 2>&1
tail -3 gpurun_out/r05e_tests_ops.log
( timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -s ) > gpurun_out/r05e_tests_e2e.log 2>&1
tail -3 gpurun_out/r05e_tests_e2e.log
( timeout 300 python tools/bench_deep.py 8 f16 ) > gpurun_out/r05e_bench_deep.txt 2>&1
tail -12 gpurun_out/r05e_bench_deep.txt
( timeout 300 python tools/launch_count.py ) > gpurun_out/r05e_launch_count.txt 2>&1
grep "^==" gpurun_out/r05e_launch_count.txt
( timeout 600 python bench.py --no-cpu-baseline ) > gpurun_out/r05e_bench.json 2> gpurun_out/r05e_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05e_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("value", d["value"]/1e6, "ms/step", d["ms_per_step"], "attn frac", r["frac"], "attn ms/fwd", r["kernel_ms_per_forward"], "fwd alone", d["roofline_forward"]["wall_ms"], "bs1", d["single_scene_latency_ms"], "paper", d["paper_protocol"]["seconds_for_312_scenes"], "bf16", d.get("bf16_head",{}).get("points_per_s"), "agree", d["agreement_vs_fp32"]["argmax_agreement"], d["agreement_vs_fp32"]["max_abs_logit_diff"])
PY
( timeout 600 python tools/host_contention.py 8 40000 8 20 ) > gpurun_out/r05e_host_contention.txt 2>&1
grep "^ranks\|^#" gpurun_out/r05e_host_contention.txt
bash tools/pmc_bench_traffic.sh $GRAFT_REPO_ROOT/gpurun_out/r05e_attention_traffic.json > gpurun_out/r05e_pmc.log 2>&1
grep "attention grid" gpurun_out/r05e_pmc.log
