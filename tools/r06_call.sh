# round 6, GPU call S: native plan builder - parity test, e2e subset, bs = 1 A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06s
( timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -k "native_plan" ) > ${O}_tests_plan.log 2>&1; tail -15 ${O}_tests_plan.log
( timeout 1200 python -m pytest tests/test_gpu_e2e.py -x -q ) > ${O}_tests_e2e.log 2>&1; tail -3 ${O}_tests_e2e.log
export CDSEG_SYNC_EACH=1
for rep in 1 2; do
  for v in base new; do
    if [ $v == base ]; then export CDSEG_AB_ENGINE=tools/_ab/engine_r06base.py; else unset CDSEG_AB_ENGINE; fi
    echo "$v:"; ( timeout 120 python tools/single_scene_profile.py ) 2>&1 | grep -v amdgpu | tail -2
  done
done | tee ${O}_bs1_host_ab.txt
unset CDSEG_AB_ENGINE
( timeout 300 python tools/host_timeline_bs1.py 30 ) 2>&1 | grep -v amdgpu > ${O}_host_timeline_bs1.txt; cat ${O}_host_timeline_bs1.txt
echo "done at $SECONDS s"
