# round 5, GPU call J: bench line with the own-process paper pass; C = 64 conv with 20 LDS-resident offsets (A/B)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r05j_conv64.txt
for rep in 1 2; do
  for lib in "" tools/_ab/libcdseg_hip_conv20.so; do
    echo "== lib=${lib:-product}" >> gpurun_out/r05j_conv64.txt
    ( CDSEG_AB_LIB=$lib CDSEG_BENCH_NEW_ONLY=1 timeout 200 python tools/bench_conv.py 1 8 30 ) 2>&1 | grep "conv level" >> gpurun_out/r05j_conv64.txt
  done
done
cat gpurun_out/r05j_conv64.txt
( timeout 900 python bench.py ) > gpurun_out/r05j_bench.json 2> gpurun_out/r05j_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05j_bench.json").read().strip().splitlines()[-1])
print("value", d["value"]/1e6, "paper", d["paper_protocol"])
PY
