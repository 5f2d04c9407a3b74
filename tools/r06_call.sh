# round 6, GPU call Z: full GPU suite on the tree (native plan with lazy views, W16 attention, cached t bias), bs = 1 timeline + A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06z
( timeout 1500 python -m pytest tests -m gpu -x -q ) > ${O}_tests.log 2>&1; tail -3 ${O}_tests.log
export CDSEG_SYNC_EACH=1
for rep in 1 2; do
  for v in base new; do
    if [ $v == base ]; then export CDSEG_AB_ENGINE=tools/_ab/engine_r06base.py; else unset CDSEG_AB_ENGINE; fi
    echo "$v:"; ( timeout 120 python tools/single_scene_profile.py ) 2>&1 | grep -v amdgpu | tail -2
  done
done | tee ${O}_bs1_host_ab.txt
unset CDSEG_AB_ENGINE
( timeout 300 python tools/host_timeline_bs1.py 14 ) 2>&1 | grep -v amdgpu > ${O}_host_timeline_bs1.txt; cat ${O}_host_timeline_bs1.txt
( timeout 300 python bench.py --protocol paper ) 2>/dev/null | tail -1 | cut -c1-300
echo "done at $SECONDS s"
