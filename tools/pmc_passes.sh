cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name, counters, cmd...
  name=$1; ctr=$2; shift 2
  rm -rf /tmp/pmc_$name
  ( cd $R && timeout -k 5 100 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_$name -o out -- "$@" > /tmp/pmc_$name.log 2>&1 )
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
    if "gemm_kernel" in k or "attn_" in k:
        key = k.split("(")[0][-60:]
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(key, r["Counter_Name"])] += 1
for k, d in acc.items():
    for c, v in d.items():
        print(f"{name} | {k} | {c} | per launch {v / cnt[(k, c)]:.0f} | launches {cnt[(k, c)]}")
PY
}
run convA "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" python tools/bench_conv.py 0 4 10
run convB "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" python tools/bench_conv.py 0 4 10
run convC "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_WAIT_INST_LDS" python tools/bench_conv.py 0 4 10
run convD "FETCH_SIZE" python tools/bench_conv.py 0 4 10
run convE "WRITE_SIZE" python tools/bench_conv.py 0 4 10
run convF "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" python tools/bench_conv.py 0 4 10
run attnD "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" python tools/bench_attention.py 120000 2 bf16 10
grep -h "conv level\|attention n=" /tmp/pmc_*.log | sort | uniq | head -4
