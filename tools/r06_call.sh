# round 6, GPU call L: fp32x3 attention block shape (8 waves x 2 tiles | 16 x 1 | 16 x 2), whole-forward A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06l
for rep in 1 2; do
  for v in product 16_1 16_2; do
    if [ $v == product ]; then unset CDSEG_AB_LIB_F16; else export CDSEG_AB_LIB_F16=tools/_ab/libcdseg_hip_f16_x3_$v.so; fi
    echo -n "$v: "; ( timeout 200 python tools/parity_mode_profile.py fp32x3 8 ) 2>&1 | grep -v amdgpu | tail -1
  done
done | tee ${O}_x3_attn_shapes.txt
unset CDSEG_AB_LIB_F16
echo "done at $SECONDS s"
