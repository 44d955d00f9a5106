ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp; cd $ROOT
O=gpurun_out/tr; rm -rf $O; mkdir -p $O
MCQUIC_AMD_BRANCH_STREAMS=0 rocprofv3 --kernel-trace -d $O/kt -o kt -- python tools/bench_train.py --steps 6 --warmup 2 > $O/bench.json 2>/dev/null
python profiles/kernel_stats.py $O/kt/kt_results.db > $O/kernel_stats_train.txt
rm -rf $O/kt
python tools/bench_train.py --graph 2>/dev/null | tail -1 > $O/bench_graph.json
