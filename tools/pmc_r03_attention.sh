# Round-3 offline PMC evidence for the attention kernel (separate --pmc passes, never with a trace domain):
#   gpurun_out/r03_attention_traffic.json  HBM bytes per launch over bench.py's own forwards (bench.py reads profiles/r03_attention_traffic.json)
#   gpurun_out/r03_attention_clock.json    the clock the kernel holds: GRBM_GUI_ACTIVE / 8 XCDs / launch duration
#   gpurun_out/pmc_r03_attn.txt            SQ / LDS / TCC counters on tools/bench_attention.py 960000 2
# usage: gpurun --timeout 1500 -- bash tools/pmc_r03_attention.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/pmc_bench_traffic.sh $GRAFT_REPO_ROOT/gpurun_out/r03_attention_traffic.json > gpurun_out/r03_pmc_traffic.log 2>&1
tail -14 gpurun_out/r03_pmc_traffic.log
bash tools/pmc_r02.sh r03_attn attn_bf16 python tools/bench_attention.py 960000 2 bf16 10 > /dev/null 2>&1
cat gpurun_out/pmc_r03_attn.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_clk
( cd $GRAFT_REPO_ROOT && timeout -k 5 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_clk -o out -- python tools/bench_attention.py 960000 2 bf16 10 > /tmp/pmc_clk.log 2>&1 )
python3 - <<'PY'
import csv, glob, json, os, re
rows = []
for f in glob.glob("/tmp/pmc_clk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_bf16" in r.get("Kernel_Name", "") and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            rows.append(r)
log = open("/tmp/pmc_clk.log").read()
m = re.search(r"median ([0-9.]+) us/launch", log)
out = {"source": "rocprofv3 --pmc GRBM_GUI_ACTIVE over tools/bench_attention.py 960000 2 bf16 (938 patches x 2 heads); "
                 "GRBM_GUI_ACTIVE is summed over the 8 XCDs", "launches": len(rows)}
if rows:
    g = sum(float(r["Counter_Value"]) for r in rows) / len(rows)
    out["GRBM_GUI_ACTIVE_per_launch"] = g
    dur = None
    if "Start_Timestamp" in rows[0] and "End_Timestamp" in rows[0]:
        try:
            dur = sum(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rows) / len(rows) * 1e-3
            out["duration_source"] = "dispatch timestamps of the same pass"
        except Exception:
            dur = None
    if not dur and m:
        dur = float(m.group(1)); out["duration_source"] = "HIP events of the same (profiled) process, median launch"
    if dur:
        out["launch_us_under_pmc"] = dur
        out["ghz"] = g / 8.0 / (dur * 1e3)
json.dump(out, open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r03_attention_clock.json"), "w"), indent=1)
print(out)
PY
