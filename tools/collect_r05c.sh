#!/bin/bash
# round 5, third collection: Neon at k = 4096 (tests, tile sweep, bench), model No. 12 measurements, full GPU suite
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
O=gpurun_out/r05c; rm -rf $O; mkdir -p $O
python -m pytest tests/test_neon.py tests/test_gpu_model12.py tests/test_gpu_step_ops.py tests/test_gpu_optim.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/summary.txt
python tools/bench_neon.py --train-batches 4,8 > $O/bench_neon.txt 2>&1
python tools/microbench_conv.py --neon --flags res > $O/neon_tile_sweep_res.txt 2>&1
python tools/microbench_conv.py --neon --flags silu_out > $O/neon_tile_sweep_silu.txt 2>&1
(time python -m pytest tests -q -m gpu -x) > $O/gputest.log 2>&1; echo "gputest rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -12 $O/tests.log; cat $O/bench_neon.txt | tail -4; cat $O/neon_tile_sweep_res.txt; tail -6 $O/gputest.log
