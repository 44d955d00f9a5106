#!/bin/bash
set -u
O=gpurun_out/r03d; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "winograd or wino" 2>&1 | tail -6 > $O/pytest.txt; tail -3 $O/pytest.txt
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], json.dumps(d['secondary']['winograd2d']))"
python tools/microbench_conv.py --winograd --big 2>&1 | tail -3
python tools/probes/graph_fork_probe.py 2>&1 | tail -5
python tools/train_conv_census.py --top 200 > $O/train_census.txt 2>&1; wc -l $O/train_census.txt
python tools/microbench_conv.py --train --flags dsilu --nprob 2 2>&1 | head -4 | cut -c1-700
python tools/microbench_conv.py --train --flags res --nprob 2 2>&1 | head -4 | cut -c1-700
python tools/microbench_conv.py --train --flags dsilu_only 2>&1 | head -4 | cut -c1-700
