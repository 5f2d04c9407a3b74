# round 5, GPU call I: attention block shape - waves per block x score tiles in flight (same-process interleaved sweep)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AB=tools/_ab
( timeout 900 python tools/attn_sweep.py --rounds 5 --knobs "0,64,64" \
   --libs exp=$AB/libcdseg_hip_exp.so,w10=$AB/libcdseg_hip_w10.so,w8t1=$AB/libcdseg_hip_w8t1.so,w10t2=$AB/libcdseg_hip_w10t2.so,w12=$AB/libcdseg_hip_w12.so \
   --shapes "864000:2:8;864000:4:8;402000:4:8;103000:8:8;27000:16:8;6200:32:8;120000:2:1;55000:4:1;14000:8:1;3400:16:1;780:32:1" ) > gpurun_out/r05i_attn_waves.txt 2>&1
grep median gpurun_out/r05i_attn_waves.txt | sed 's/checksum/chk/' | cut -c1-150
