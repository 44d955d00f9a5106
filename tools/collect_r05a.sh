#!/bin/bash
# round 5, first collection: new step-op tests, autograd fan-in sites, captured-step sequence, full GPU suite
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
O=gpurun_out/r05a; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_step_ops.py -x -q -m gpu > $O/step_ops.log 2>&1; echo "step_ops rc=$?" >> $O/summary.txt
python tools/autograd_fanin.py > $O/fanin.txt 2>&1
python tools/bench_train.py --graph --steps 20 2>/dev/null | tail -1 > $O/bench_graph.json
python tools/bench_train.py --graphed --steps 20 2>/dev/null | tail -1 > $O/bench_graphed.json
rocprofv3 --kernel-trace -d $O/kt -o kt -- python tools/bench_train.py --graph --steps 6 > $O/kt.log 2>&1
python tools/probes/dump_step_sequence.py $O/kt/kt_results.db > $O/train_step_sequence.txt 2>&1
rm -rf $O/kt
(time python -m pytest tests -x -q -m gpu) > $O/gputest.log 2>&1; echo "gputest rc=$?" >> $O/summary.txt
cat $O/summary.txt; cat $O/bench_graph.json $O/bench_graphed.json; tail -5 $O/step_ops.log; head -3 $O/train_step_sequence.txt; tail -8 $O/gputest.log
