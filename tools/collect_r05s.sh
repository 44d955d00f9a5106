# round 5: which tile for the single-round 128x128 launches of the training step, per epilogue (isolated; first column carries the clock ramp)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s; rm -rf $O; mkdir -p $O
for fl in res dual dsilu dsilu_only silu_out resonly; do
  echo "== $fl" >> $O/micro.txt
  timeout 300 python tools/microbench_conv.py --train --flags $fl --tiles 0x11,0x42,0x41,0x22,0x21,0x442 2>/dev/null | grep -v "^lib" | head -1 | cut -d'|' -f2- >> $O/micro.txt
  timeout 300 python tools/microbench_conv.py --train --nprob 2 --flags $fl --tiles 0x11,0x42,0x41,0x22,0x21,0x442 2>/dev/null | grep -v "^lib" | head -2 | tail -1 | cut -d'|' -f2- >> $O/micro.txt
done
cat $O/micro.txt
