# round 5, call N (the last GPU minutes of the round): does the LANE MAP of the gathered-row loads bound the wide-stage sparse
# conv?  (1) bare loads, both maps (tools/ubench/gather_map); (2) same-process sweep of conv.hip builds (tools/conv_sweep.py:
# base, nb = deeper row ring, quad = quad-contiguous loads + ds_bpermute, quadnb = both), bit-for-bit against base in both
# 16-bit builds; (3) if a variant is bit-identical and >= 3 % faster on the forward's conv mix, the WHOLE GPU suite + smoke run
# on that variant's library pair (then it may become the default), and a same-box end-to-end A/B if time is left;
# otherwise counters of the base kernel's vector-memory path.  Self-limited: the call has ~10 GPU minutes.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r05n
LIMIT=${R05N_LIMIT:-585}
left() { echo $(( LIMIT - SECONDS )); }
( timeout 90 tools/ubench/gather_map ) > ${O}_gather_map.txt 2>&1; tail -4 ${O}_gather_map.txt
( timeout 240 python tools/conv_sweep.py --libs base=,nb=tools/_ab/nb,quad=tools/_ab/quad,quad64=tools/_ab/quad64,quadnb=tools/_ab/quadnb,quadcf=tools/_ab/quadcf,quadall=tools/_ab/quadall --out ${O}_sweep.json ) > ${O}_sweep.txt 2>&1
cat ${O}_sweep.txt | tail -12
W=$(python3 - <<'PY'
import json
try:
    r = json.load(open("gpurun_out/r05n_sweep.json"))
except Exception:
    print("base"); raise SystemExit
cost = lambda u: 4 * u["32"] + 8 * u["64"]  # the forward's 12 wide convs: 4 at C = 32, 8 at C = 64
base = cost(r["us"]["base"])
best, bc = "base", base
for name, u in r["us"].items():
    if name != "base" and r["equal"].get(name) and cost(u) < 0.97 * base and cost(u) < bc:
        best, bc = name, cost(u)
print(best)
PY
)
echo "winner: $W ($(left) s left)" | tee ${O}_choice.txt
if [ "$W" != "base" ]; then
  cp cdsegnet_amd/libcdseg_hip.so /tmp/keep_lib.so; cp cdsegnet_amd/libcdseg_hip_f16.so /tmp/keep_lib_f16.so
  cp tools/_ab/$W/libcdseg_hip.so tools/_ab/$W/libcdseg_hip_f16.so cdsegnet_amd/
  touch cdsegnet_amd/libcdseg_hip.so cdsegnet_amd/libcdseg_hip_f16.so
  T=$(left); [ $T -gt 30 ] && ( timeout $T python -m pytest tests -m gpu -x -q ) > ${O}_tests.log 2>&1
  tail -3 ${O}_tests.log | tee -a ${O}_choice.txt
  for v in $W base; do
    T=$(left); [ $T -lt 50 ] && break
    if [ $v == base ]; then cp /tmp/keep_lib.so cdsegnet_amd/libcdseg_hip.so; cp /tmp/keep_lib_f16.so cdsegnet_amd/libcdseg_hip_f16.so
    else cp tools/_ab/$v/libcdseg_hip.so tools/_ab/$v/libcdseg_hip_f16.so cdsegnet_amd/; fi
    touch cdsegnet_amd/libcdseg_hip.so cdsegnet_amd/libcdseg_hip_f16.so
    timeout $T python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-agreement --no-paper-pass --no-kernel-timer 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('e2e $v', round(d['value']/1e6,2), 'M points/s', round(d['ms_per_step'],2), 'ms/step')" | tee -a ${O}_choice.txt
  done
  cp tools/_ab/$W/libcdseg_hip.so tools/_ab/$W/libcdseg_hip_f16.so cdsegnet_amd/; touch cdsegnet_amd/libcdseg_hip.so cdsegnet_amd/libcdseg_hip_f16.so
  T=$(left); [ $T -gt 35 ] && ( timeout $T python -c "import __graft_entry__ as g; g.smoke()" ) > ${O}_smoke.log 2>&1; tail -2 ${O}_smoke.log | tee -a ${O}_choice.txt
else
  # the base kernel's vector-memory path in counters (own passes, no trace domains)
  cd /tmp && export TMPDIR=/tmp
  R=$GRAFT_REPO_ROOT
  run() { # name, counters
    T=$(( LIMIT - SECONDS )); [ $T -lt 45 ] && return
    rm -rf /tmp/pmc_$1
    ( cd $R && CDSEG_BENCH_NEW_ONLY=1 timeout -k 5 $T rocprofv3 --pmc $2 --output-format csv -d /tmp/pmc_$1 -o out -- python tools/bench_conv.py 1 8 10 > /tmp/pmc_$1.log 2>&1 )
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
    run A "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum"
    run B "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
    run C "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
    run D "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TD_TD_BUSY_sum TD_TC_STALL_sum"
    run E "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
  } > $R/${O}_pmc_conv64.txt 2>&1
  tail -30 $R/${O}_pmc_conv64.txt
fi
echo "done at $SECONDS s"
