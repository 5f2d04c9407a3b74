# round 5, GPU call F: same-box A/B of the round's two kernel changes (tail zones of the attention schedule; C = 512 fused
# head / tail), executor tests with the row threshold, bs = 1 census / latency
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "deep_head or block_executor" ) > gpurun_out/r05f_tests_ops.log 2>&1
tail -2 gpurun_out/r05f_tests_ops.log
( bash tools/ab_value.sh 3 on nozones nodeep512 ) > gpurun_out/r05f_ab.txt 2>&1
cat gpurun_out/r05f_ab.txt
( timeout 300 python tools/launch_count.py ) > gpurun_out/r05f_launch_count.txt 2>&1
grep "^==" gpurun_out/r05f_launch_count.txt
( timeout 300 python tools/single_scene_profile.py ) > gpurun_out/r05f_single.txt 2>&1
tail -3 gpurun_out/r05f_single.txt
( timeout 300 python bench.py --protocol paper ) > gpurun_out/r05f_paper.json 2>/dev/null
cut -c1-200 gpurun_out/r05f_paper.json
