# round 5, GPU call A: correctness of the reworked attention path, schedule sweep, launch / copy census, first bench.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "attention or block_executor or deep_head" ) > gpurun_out/r05a_tests_attn.log 2>&1
tail -5 gpurun_out/r05a_tests_attn.log
( timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q ) > gpurun_out/r05a_tests_e2e.log 2>&1
tail -5 gpurun_out/r05a_tests_e2e.log
AB=tools/_ab
( timeout 500 python tools/attn_sweep.py --libs base=$AB/libcdseg_hip_base.so,exp=$AB/libcdseg_hip_exp.so,ns16=$AB/libcdseg_hip_exp_ns16.so \
   --knobs "0,0,0;0,0,64;0,64,64;0,64,128;0,128,128;32,0,0;32,64,64;64,64,128;0,32,32;0,0,128" ) > gpurun_out/r05a_attn_sweep.txt 2>&1
tail -80 gpurun_out/r05a_attn_sweep.txt
( timeout 300 python tools/launch_count.py ) > gpurun_out/r05a_launch_count.txt 2>&1
head -12 gpurun_out/r05a_launch_count.txt
( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err
head -c 400 gpurun_out/r05a_bench.json; echo
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05a_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("value", d["value"]/1e6, "ms/step", d["ms_per_step"], "attn frac", r["frac"], "attn ms/fwd", r["kernel_ms_per_forward"], "fwd alone", d["roofline_forward"]["wall_ms"], "bs1", d["single_scene_latency_ms"], "agree", d["agreement_vs_fp32"]["argmax_agreement"], d["agreement_vs_fp32"]["max_abs_logit_diff"])
PY
