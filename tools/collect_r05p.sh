# round 5: pair tile -- parity of all its epilogues, isolated training shapes, and the captured training step with the automatic rule
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05p; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "pair" > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt
for fl in res dual dsilu dsilu_only; do
  echo "== train $fl" >> $O/micro.txt
  timeout 300 python tools/microbench_conv.py --train --flags $fl --tiles 0x42,0x41,0x442 2>/dev/null | grep -v "^lib" | head -1 >> $O/micro.txt
done
export MCQUIC_AMD_LIB=$PWD/mcquic_amd/variants/libmcquic_pair.so
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_graphed_step.py -q -m gpu -x > $O/tests_pairlib.log 2>&1; echo "pairlib tests rc=$?" >> $O/summary.txt
for i in 1 2 3; do for v in main pair; do
  if [ $v = main ]; then unset MCQUIC_AMD_LIB; else export MCQUIC_AMD_LIB=$PWD/mcquic_amd/variants/libmcquic_$v.so; fi
  timeout 600 python tools/bench_train.py --graph --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])" >> $O/ab.txt
done; done
tail -5 $O/tests.log; tail -5 $O/tests_pairlib.log; cat $O/summary.txt $O/micro.txt $O/ab.txt
