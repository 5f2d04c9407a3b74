# round-6 validation on the GPU box: full GPU test suite (with the [measure] lines), smoke, PMC traffic pass, the default
# bench line, kernel-trace profiles (default and one-lane serial), the other BASELINE configurations.
# usage: gpurun --timeout 2700 -- bash tools/r06_final.sh   (outputs under gpurun_out/r06g_*, copied to profiles/r06_* by hand)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q -s ) > gpurun_out/r06g_tests.log 2>&1
tail -3 gpurun_out/r06g_tests.log
grep "\[measure\]" gpurun_out/r06g_tests.log | sed 's/^\.*//' > gpurun_out/r06g_parity_measured.txt; wc -l gpurun_out/r06g_parity_measured.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r06g_smoke.log 2>&1; tail -4 gpurun_out/r06g_smoke.log
bash tools/pmc_bench_traffic.sh $GRAFT_REPO_ROOT/gpurun_out/r06g_attention_traffic.json > gpurun_out/r06g_pmc.log 2>&1
tail -14 gpurun_out/r06g_pmc.log
cp gpurun_out/r06g_attention_traffic.json profiles/r06_attention_traffic.json
cd $GRAFT_REPO_ROOT
( timeout 900 python bench.py ) > gpurun_out/r06g_bench.json 2> gpurun_out/r06g_bench.err
head -c 600 gpurun_out/r06g_bench.json; echo
cd /tmp && export TMPDIR=/tmp
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_final -o run -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-agreement --no-kernel-timer --no-paper-pass > gpurun_out/r06g_bench_under_rocprof.json 2> gpurun_out/r06g_prof.err )
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_final_l1 -o run -- python bench.py --scenes-per-forward 8 --lanes 1 --serial --steps 6 --warmup 2 --no-cpu-baseline --no-agreement --no-kernel-timer --no-paper-pass > gpurun_out/r06g_bench_lanes1.json 2>> gpurun_out/r06g_prof.err )
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_final_l1b -o run -- python bench.py --lanes 1 --serial --steps 4 --warmup 2 --no-cpu-baseline --no-agreement --no-kernel-timer --no-paper-pass > gpurun_out/r06g_bench_lanes1_24.json 2>> gpurun_out/r06g_prof.err )
( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_final1 -o run -- python tools/single_scene_profile.py > gpurun_out/r06g_single.txt 2> gpurun_out/r06g_single.err )
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_final -name "*.db" | head -1); python tools/prof_summary.py $DB 6 > gpurun_out/r06g_kernel_stats.txt 2>&1
DB=$(find /tmp/prof_final_l1 -name "*.db" | head -1); python tools/prof_summary.py $DB 8 > gpurun_out/r06g_lanes1_kernel_stats.txt 2>&1
DB=$(find /tmp/prof_final_l1b -name "*.db" | head -1); python tools/prof_summary.py $DB 6 > gpurun_out/r06g_lanes1_24_kernel_stats.txt 2>&1
DB=$(find /tmp/prof_final1 -name "*.db" | head -1); python tools/prof_summary.py $DB 30 > gpurun_out/r06g_single_kernel_stats.txt 2>&1
cat gpurun_out/r06g_single.txt; head -30 gpurun_out/r06g_lanes1_kernel_stats.txt | cut -c1-150
: > gpurun_out/r06g_other_configs.txt
other() {
  ( timeout 300 python bench.py --no-cpu-baseline --no-paper-pass --steps 8 "$@" ) > /tmp/other.json 2> /tmp/other.err
  python - "$*" >> gpurun_out/r06g_other_configs.txt <<'PY'
import json, sys
try:
    d = json.loads(open("/tmp/other.json").read().strip().splitlines()[-1])
    r = d.get("roofline"); a = d.get("agreement_vs_fp32") or {}
    if r is None:
        print(f"bench.py {sys.argv[1]} : {json.dumps(d)}")
    else:
        print(f"bench.py {sys.argv[1]} : {d['value'] / 1e6:.2f} M points/s, {d['ms_per_step']:.2f} ms/step, "
              f"{d['config']['scenes_per_step_per_gpu']} scenes/step, mean points/scene {d['config']['points_per_scene_mean']:.0f}, "
              f"attention frac {r['frac']:.4f}, forward alone {d['roofline_forward']['wall_ms']:.2f} ms, bs=1 {d['single_scene_latency_ms']:.2f} ms, "
              f"agreement vs fp32 {a.get('argmax_agreement')}")
except Exception as e:  # noqa: BLE001
    print(f"bench.py {sys.argv[1]} : failed ({e}); stderr tail: {open('/tmp/other.err').read()[-400:]}")
PY
}
other --dataset scannet200
other --dataset nuscenes --points 40000
other --robust
other --precision bf16+head
other --precision fp32 --scenes-per-forward 8 --lanes 2
other --scenes-per-forward 8 --lanes 3
other --protocol paper
other --dataset nuscenes --points 40000 --shard 64
cat gpurun_out/r06g_other_configs.txt
# attention counters (separate --pmc passes) with and without the producer-side preprocessing flag, and the CPU baseline's
# thread sweep at the baseline's own scene size
bash tools/pmc_r02.sh r06_attn attn_bf16 python tools/bench_attention.py 960000 2 bf16 10 2 1 3 > /dev/null 2>&1
cat gpurun_out/pmc_r06_attn.txt | head -30
( timeout 400 python tools/cpu_sweep.py 120000 8 16 32 ) > gpurun_out/r06g_cpu_sweep.txt 2>&1
cat gpurun_out/r06g_cpu_sweep.txt
