# round 6, GPU call A: (1) tools/ubench/gather_map incl. the buffer-load / dead-row / 89 %-hit case and the whole-row + LDS-DMA
# maps; (2) counters of conv_ll_kernel<64> in passes of <= 3; (3) baseline bench of the round's starting tree; (4) bs = 1 wall.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06a
( timeout 150 tools/ubench/gather_map ) > ${O}_gather_map.txt 2>&1; tail -30 ${O}_gather_map.txt
( timeout 300 python bench.py ) > ${O}_bench.json 2> ${O}_bench.err; tail -c 600 ${O}_bench.json
( timeout 200 python tools/single_scene_profile.py ) > ${O}_single.txt 2>&1; tail -2 ${O}_single.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name, counters
  rm -rf /tmp/pmc_$1
  ( cd $R && CDSEG_BENCH_NEW_ONLY=1 timeout -k 5 90 rocprofv3 --pmc $2 --output-format csv -d /tmp/pmc_$1 -o out -- python tools/bench_conv.py 1 8 10 > /tmp/pmc_$1.log 2>&1 )
  f=$(find /tmp/pmc_$1 -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$1" <<'PY'
import csv, sys, collections
f, name = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); cnt = collections.Counter()
try:
    rows = list(csv.DictReader(open(f)))
except Exception as e:
    print(name, "no csv", e); sys.exit(0)
for r in rows:
    if "conv_ll_kernel" in r.get("Kernel_Name", ""):
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for c, v in acc.items():
    print(f"{name} | conv_ll_kernel<64> | {c} | per launch {v / cnt[c]:.0f} | launches {cnt[c]}")
PY
  grep -h "conv level" /tmp/pmc_$1.log | tail -1
}
{
  run A "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum"
  run B "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
  run C "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
  run D "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TD_TD_BUSY_sum"
  run E "SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
  run F "TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
} > $R/${O}_pmc_conv64.txt 2>&1
tail -30 $R/${O}_pmc_conv64.txt
echo "done at $SECONDS s"
