# round 5: the 128 x 64 tile over pixel PAIRS (tile bit 0x400) against the standing tile, parity + isolated launches
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "pair" > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt
for fl in silu_out res dual resonly; do
  echo "== $fl" >> $O/micro.txt
  timeout 300 python tools/microbench_conv.py --big --flags $fl --tiles 0x42,0x442 2>/dev/null | grep -v "^lib" >> $O/micro.txt
  timeout 300 python tools/microbench_conv.py --train --flags $fl --tiles 0x42,0x442 2>/dev/null | grep -v "^lib" | head -2 >> $O/micro.txt
done
tail -15 $O/tests.log; cat $O/summary.txt $O/micro.txt
