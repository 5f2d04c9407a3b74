cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06af
t0=$SECONDS
( timeout 900 python bench.py ) > ${O}_bench.json 2> ${O}_bench.err
echo "bench.py took $((SECONDS - t0)) s"
python3 -c "
import json
d=json.loads(open('${O}_bench.json').read().strip().splitlines()[-1])
print('e2e', round(d['value']/1e6,2), 'M; ms/step', round(d['ms_per_step'],2), 'attn frac', round(d['roofline']['frac'],4), 'bs1', d['headline']['bs1_ms_per_scene'], 'host_issue', d['host_issue']['host_issue_ms_per_forward'], 'fwd alone', d['roofline_forward']['wall_ms'], 'single', d['single_scene_latency_ms'])
print('r5cfg', d.get('value_at_round5_config'))
print('parity', {k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if a != 'note'}) for k, v in d['parity_mode'].items() if k != 'note'})
print('bf16', d['bf16_head']['points_per_s'], 'paper', d['paper_protocol']['seconds_for_312_scenes'], d['paper_protocol'].get('own_process'))
print('conv', d['roofline_conv']['frac'], d['roofline_conv']['avg_launch_us'], 'deep', d['roofline_conv_deep']['frac'], d['roofline_conv_deep']['avg_launch_us'], 'stale', d['roofline'].get('traffic_stale'))
"
tail -2 ${O}_bench.err
echo "done at $SECONDS s"
