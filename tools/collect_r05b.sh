#!/bin/bash
# round 5, second collection: step-op / optimizer / model12 tests, captured-step sequence, bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
O=gpurun_out/r05b; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_step_ops.py tests/test_gpu_optim.py tests/test_gpu_model12.py tests/test_gpu_backward.py tests/test_gpu_train_forward.py tests/test_gpu_graphed_step.py tests/test_gpu_golden.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/summary.txt
python tools/bench_train.py --graph --steps 20 2>/dev/null | tail -1 > $O/bench_graph.json
python tools/bench_train.py --graphed --steps 20 2>/dev/null | tail -1 > $O/bench_graphed.json
rocprofv3 --kernel-trace -d $O/kt -o kt -- python tools/bench_train.py --graph --steps 6 > $O/kt.log 2>&1
python tools/probes/dump_step_sequence.py $O/kt/kt_results.db > $O/train_step_sequence.txt 2>&1
rm -rf $O/kt
python bench.py > $O/bench.json 2> $O/bench.err
cat $O/summary.txt; cat $O/bench_graph.json $O/bench_graphed.json; tail -15 $O/tests.log; head -2 $O/train_step_sequence.txt; tail -c 3000 $O/bench.json
