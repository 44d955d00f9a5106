#!/bin/bash
# round 5, fourth collection: block nodes + shuffle side tensors + narrow-layer tile rules: tests, A/B of the dominant instance,
# training step, Neon bench, batch-1 trace
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
O=gpurun_out/r05d; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_step_ops.py tests/test_neon.py tests/test_gpu_model12.py tests/test_gpu_backward.py tests/test_gpu_graphed_step.py tests/test_gpu_golden.py tests/test_gpu_ops.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/summary.txt
# A/B: the headline with / without the side tensors in the PixelShuffle store (register allocation of the dominant instance)
for i in 1 2; do
  python bench.py --no-secondary --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('main', d['value'], d['ms_per_step'], d['roofline']['frac'])" >> $O/ab_headline.txt
  MCQUIC_AMD_LIB=$ROOT/mcquic_amd/variants/libmcquic_noshufside.so MCQUIC_AMD_BLOCK_NODES=0 python bench.py --no-secondary --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('noshufside', d['value'], d['ms_per_step'], d['roofline']['frac'])" >> $O/ab_headline.txt
done
for i in 1 2; do
  python tools/bench_train.py --graph --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('main', d['ms_per_step'])" >> $O/ab_train.txt
  MCQUIC_AMD_BLOCK_NODES=0 python tools/bench_train.py --graph --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('opbyop', d['ms_per_step'])" >> $O/ab_train.txt
  MCQUIC_AMD_LIB=$ROOT/mcquic_amd/variants/libmcquic_noshufside.so MCQUIC_AMD_BLOCK_NODES=0 python tools/bench_train.py --graph --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('noshufside+opbyop', d['ms_per_step'])" >> $O/ab_train.txt
done
python tools/bench_train.py --graphed --steps 20 2>/dev/null | tail -1 > $O/bench_graphed.json
rocprofv3 --kernel-trace -d $O/kt -o kt -- python tools/bench_train.py --graph --steps 6 > $O/kt.log 2>&1
python tools/probes/dump_step_sequence.py $O/kt/kt_results.db > $O/train_step_sequence.txt 2>&1
rm -rf $O/kt
python tools/bench_neon.py --train-batches 4,8 > $O/bench_neon.txt 2>&1
python tools/microbench_conv.py --neon --flags res --tiles 0 > $O/neon_auto_tiles.txt 2>&1
rocprofv3 --kernel-trace -d $O/ktb1 -o kt -- python bench.py --batch 1 --graphs --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_b1.json 2>/dev/null
python profiles/kernel_stats.py $O/ktb1/kt_results.db > $O/kernel_stats_b1.txt
python tools/probes/dump_step_sequence.py $O/ktb1/kt_results.db > $O/b1_sequence.txt 2>&1
rm -rf $O/ktb1
cat $O/summary.txt; tail -8 $O/tests.log; cat $O/ab_headline.txt $O/ab_train.txt; cat $O/bench_graphed.json; head -2 $O/train_step_sequence.txt; tail -5 $O/bench_neon.txt; cat $O/neon_auto_tiles.txt; head -3 $O/b1_sequence.txt
