# round 5, GPU call C: interleaved schedule sweep (patch-head zones), tests of the product build, bench with tentative defaults
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AB=tools/_ab
( timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "attention" ) > gpurun_out/r05c_tests_attn.log 2>&1
tail -3 gpurun_out/r05c_tests_attn.log
KN="0,0,0;0,32,32;0,64,64;0,32,64;0,0,64;0,0,128;0,64,0;32,32,32;32,0,64;0,16,32;0,128,128"
( timeout 900 python tools/attn_sweep.py --rounds 5 --libs base=$AB/libcdseg_hip_base.so,exp=$AB/libcdseg_hip_exp.so --knobs "$KN" \
   --shapes "864000:2:8;864000:4:8;402000:4:8;103000:8:8;27000:16:8;6200:32:8;120000:2:1;55000:4:1;14000:8:1;3400:16:1;780:32:1" ) > gpurun_out/r05c_attn_sweep.txt 2>&1
grep -c median gpurun_out/r05c_attn_sweep.txt
( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > gpurun_out/r05c_bench.json 2> gpurun_out/r05c_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05c_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("value", d["value"]/1e6, "ms/step", d["ms_per_step"], "attn frac", r["frac"], "attn ms/fwd", r["kernel_ms_per_forward"], "fwd alone", d["roofline_forward"]["wall_ms"], "bs1", d["single_scene_latency_ms"], "agree", d["agreement_vs_fp32"]["argmax_agreement"], d["agreement_vs_fp32"]["max_abs_logit_diff"])
PY
