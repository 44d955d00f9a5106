#!/bin/bash
# kernel trace of the opt-in Winograd bench (gpurun, from the repo root) -> gpurun_out/wino/
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp; cd $ROOT
O=gpurun_out/wino; rm -rf $O; mkdir -p $O
MCQUIC_AMD_BRANCH_STREAMS=0 rocprofv3 --kernel-trace -d $O/kt -o kt -- python bench.py --winograd ${WINO:-1} --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_ss.json 2>/dev/null
python profiles/kernel_stats.py $O/kt/kt_results.db > $O/kernel_stats_winograd_single_stream.txt
rm -rf $O/kt
