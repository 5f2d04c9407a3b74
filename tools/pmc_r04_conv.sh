# PMC passes over the deep sparse convs (tools/bench_conv.py, 8 scenes): MFMA instructions issued vs the occupied minimum,
# LDS bank conflicts, HBM bytes.  Counters in their own runs (no trace domains).  usage: gpurun -- bash tools/pmc_r04_conv.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
run() { # name, counters, level
  name=$1; ctr=$2; lvl=$3
  rm -rf /tmp/pmc_$name
  ( cd $R && CDSEG_BENCH_OLD_ONLY=1 timeout -k 5 120 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_$name -o out -- python tools/bench_conv.py $lvl 8 10 > /tmp/pmc_$name.log 2>&1 )
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$name" <<'PY'
import csv, sys, collections
f, name = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
try:
    rows = list(csv.DictReader(open(f)))
except Exception as e:
    print(name, "no csv", e); sys.exit(0)
for r in rows:
    k = r.get("Kernel_Name", "")
    if "gemm_dma_kernel" in k or "splitk" in k:
        key = k.split("(")[0][-48:]
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(key, r["Counter_Name"])] += 1
for k, d in acc.items():
    for c, v in d.items():
        print(f"{name} | {k} | {c} | per launch {v / cnt[(k, c)]:.0f} | launches {cnt[(k, c)]}")
PY
  grep -h "conv level" /tmp/pmc_$name.log | tail -1
}
for lvl in 2 3 4; do
  run c${lvl}A "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD" $lvl
  run c${lvl}B "FETCH_SIZE" $lvl
  run c${lvl}C "WRITE_SIZE" $lvl
done
