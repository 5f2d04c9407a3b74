# rocprofv3 --pmc passes (one counter set per run; never combined with trace domains).  The last lines of each block are the
# tool's own timing output UNDER the profiler: the duration the counters of that pass belong to (PMC passes run the kernel
# 1.05-4x slower than unprofiled; the unprofiled duration is in the bench line / the A/B files).
# usage: bash tools/pmc_r02.sh <tag> <kernel substring> <cmd...>   -> gpurun_out/pmc_<tag>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; kern=$2; shift 2
out=$R/gpurun_out/pmc_$tag.txt
: > $out
run() { # name, counters
  name=$1; ctr=$2
  rm -rf /tmp/pmc_$name
  ( cd $R && timeout -k 5 150 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_$name -o out -- "${CMD[@]}" > /tmp/pmc_$name.log 2>&1 )
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$name" "$kern" >> $out <<'PY'
import csv, sys, collections
f, name, kern = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
try:
    rows = list(csv.DictReader(open(f)))
except Exception as e:
    print(name, "no csv", e); sys.exit(0)
for r in rows:
    k = r.get("Kernel_Name", "")
    if kern in k:
        key = k.split("(")[0][-50:]
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(key, r["Counter_Name"])] += 1
for k, d in acc.items():
    for c, v in d.items():
        print(f"{name} | {k} | {c} | per launch {v / cnt[(k, c)]:.0f} | launches {cnt[(k, c)]}")
PY
  grep -h "conv level\|attention\[\|stem5 n=" /tmp/pmc_$name.log | tail -2 >> $out
}
CMD=("$@")
run ${tag}A "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES"
run ${tag}B "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES"
run ${tag}C "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
run ${tag}D "FETCH_SIZE"
run ${tag}E "WRITE_SIZE"
run ${tag}F "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
cat $out
