#!/bin/bash
set -u
O=gpurun_out/r03e; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_backward.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -6 > $O/pytest.txt; tail -3 $O/pytest.txt
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], json.dumps(d['secondary']))"
python tools/bench_train.py --graph --steps 10 2>/dev/null | tail -1
python tools/microbench_conv.py --train --flags dsilu_only 2>&1 | head -4 | cut -c1-900
python tools/microbench_conv.py --train --flags res 2>&1 | sed -n 3,3p | cut -c1-900
python tools/train_conv_census.py --top 200 > $O/train_census.txt 2>&1; head -30 $O/train_census.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MCQUIC_AMD_BRANCH_STREAMS=0 rocprofv3 --kernel-trace -d $O/kt -o kt -- python tools/bench_train.py --steps 6 --warmup 2 > /dev/null 2>&1
python profiles/kernel_stats.py $O/kt/kt_results.db > $O/kernel_stats_train.txt; rm -rf $O/kt; head -40 $O/kernel_stats_train.txt | cut -c1-170
