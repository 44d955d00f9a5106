#!/bin/bash
# Round-3 exploration run: lockstep / soft-assign validation + where the training step's time goes.
set -u
O=gpurun_out/r03c; mkdir -p $O
python -m pytest tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_train_forward.py tests/test_gpu_backward.py tests/test_neon.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -15 > $O/pytest.txt
tail -4 $O/pytest.txt
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print(json.dumps(d['secondary']))"
MCQUIC_AMD_MULTI_MAX_PIXELS=0 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | cut -c1-300
python bench.py --batch 1 --graphs --steps 50 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | cut -c1-260
python tools/train_conv_census.py > $O/train_census.txt 2>&1; head -45 $O/train_census.txt
python tools/train_conv_census.py --eval --batch 32 --height 768 --width 512 --iters 10 > $O/infer_census.txt 2>&1; head -30 $O/infer_census.txt
python tools/microbench_conv.py --train --flags res > $O/mb_train_res.txt 2>&1; cat $O/mb_train_res.txt
python tools/microbench_conv.py --train --flags dsilu > $O/mb_train_dsilu.txt 2>&1; cat $O/mb_train_dsilu.txt
