# round 6, GPU call C: whole GPU suite on the new attention kernel, same-box end-to-end A/B against the round-5 kernel, the new
# bench line (ceiling, conv classes, headline block)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06c
( timeout 900 python -m pytest tests -m gpu -x -q ) > ${O}_tests.log 2>&1; tail -3 ${O}_tests.log
( bash tools/ab_value.sh 2 oldattn new ) > ${O}_ab.txt 2>&1; cat ${O}_ab.txt
( timeout 400 python bench.py --no-cpu-baseline ) > ${O}_bench.json 2> ${O}_bench.err; tail -c 1500 ${O}_bench.json; tail -3 ${O}_bench.err
echo "done at $SECONDS s"
