# round 5: stride-2 input-gradient launches walk 4 of 9 taps (MCQ_CONV_TAPS_LR) -- parity, gradients, captured training step A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05q; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "four_taps or pair" > $O/tests.log 2>&1; echo "ops tests rc=$?" > $O/summary.txt
timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_graphed_step.py tests/test_gpu_train_forward.py -q -m gpu -x > $O/tests_bwd.log 2>&1; echo "backward tests rc=$?" >> $O/summary.txt
for i in 1 2 3; do
  MCQUIC_AMD_TAPS_LR=0 timeout 600 python tools/bench_train.py --graph --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nine_taps', d['ms_per_step'])" >> $O/ab.txt
  timeout 600 python tools/bench_train.py --graph --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('four_taps', d['ms_per_step'])" >> $O/ab.txt
done
tail -4 $O/tests.log; tail -3 $O/tests_bwd.log; cat $O/summary.txt $O/ab.txt
