cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_step_ops.py tests/test_gpu_graphed_step.py tests/test_gpu_graph_replay.py tests/test_host_abi.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt
for i in 1 2; do
python tools/bench_train.py --graph --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('defer', d['ms_per_step'])" >> $O/ab.txt
MCQUIC_AMD_WGRAD_DEFER=0 python tools/bench_train.py --graph --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nodefer', d['ms_per_step'])" >> $O/ab.txt
done
rocprofv3 --kernel-trace -d $O/kt -o kt -- python tools/bench_train.py --graph --steps 6 > $O/kt.log 2>&1
python tools/probes/dump_step_sequence.py $O/kt/kt_results.db > $O/train_step_sequence.txt 2>&1
rm -rf $O/kt
tail -5 $O/tests.log; cat $O/summary.txt $O/ab.txt; head -1 $O/train_step_sequence.txt; python tools/step_buckets.py $O/train_step_sequence.txt
