# HBM traffic of bench.py's own forwards, per kernel, from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
# never combined with a trace domain).  usage: bash tools/pmc_bench_traffic.sh <out.json>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=${1:-$R/gpurun_out/r03_attention_traffic.json}
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$ctr
  ( cd $R && timeout -k 5 400 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmcb_$ctr -o out -- \
      python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-agreement --no-kernel-timer > /tmp/pmcb_$ctr.log 2>&1 )
  tail -2 /tmp/pmcb_$ctr.log | cut -c1-300
done
python3 - "$out" <<'PY'
import csv, glob, hashlib, json, os, sys, collections  # noqa: E401
out = sys.argv[1]
tot = {}
by_grid = {}  # attention launches by grid size (= by stage shape): counter -> grid -> [sum, launches]
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    acc, cnt = collections.Counter(), collections.Counter()
    for f in glob.glob(f"/tmp/pmcb_{ctr}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].strip()
            acc[k] += float(r["Counter_Value"]); cnt[k] += 1
            if k.startswith("attn_bf16_kernel") and "Grid_Size" in r:
                e = by_grid.setdefault(ctr, {}).setdefault(int(r["Grid_Size"]), [0.0, 0])
                e[0] += float(r["Counter_Value"]); e[1] += 1
    tot[ctr] = (acc, cnt)
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 0` "
                 "(3 forwards of 8 collated scenes); KB = 1024 B; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM)",
       "kernels": {}}
for k in tot["FETCH_SIZE"][0]:
    n = tot["FETCH_SIZE"][1][k]
    f_kb = tot["FETCH_SIZE"][0][k] / n
    w_kb = tot["WRITE_SIZE"][0].get(k, 0.0) / max(1, tot["WRITE_SIZE"][1].get(k, 0))
    res["kernels"][k] = {"launches": n, "FETCH_SIZE_KB_raw_per_launch": f_kb, "WRITE_SIZE_KB_per_launch": w_kb,
                         "hbm_bytes_per_launch": (2 * f_kb + w_kb) * 1024}
a = [v for k, v in res["kernels"].items() if k.startswith("attn_bf16_kernel")]
if a:
    res["kernel"] = "attn_bf16_kernel"
    res.update({k: a[0][k] for k in a[0]})
res["attention_by_grid"] = {
    str(g): {"launches": by_grid["FETCH_SIZE"][g][1],
             "read_MB_per_launch": 2 * by_grid["FETCH_SIZE"][g][0] / by_grid["FETCH_SIZE"][g][1] * 1024 / 1e6,
             "write_MB_per_launch": by_grid.get("WRITE_SIZE", {}).get(g, [0.0, 1])[0] / max(1, by_grid.get("WRITE_SIZE", {}).get(g, [0.0, 1])[1]) * 1024 / 1e6}
    for g in sorted(by_grid.get("FETCH_SIZE", {}))}
# the counters belong to the kernel source they were taken on: bench.py marks them stale when attention.hip has changed since
src = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "cdsegnet_amd", "csrc", "attention.hip")
if os.path.exists(src):
    res["attention_hip_sha256"] = hashlib.sha256(open(src, "rb").read()).hexdigest()
json.dump(res, open(out, "w"), indent=1)
for g, v in res["attention_by_grid"].items():
    print(f"attention grid {g:>9s} threads: launches {v['launches']:3d}  read {v['read_MB_per_launch']:8.2f} MB  write {v['write_MB_per_launch']:8.2f} MB per launch")
big = sorted(res["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]
for k, v in big:
    print(f"{k[:60]:60s} launches {v['launches']:5d}  MB/launch {v['hbm_bytes_per_launch'] / 1e6:9.2f}")
PY
