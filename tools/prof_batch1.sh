ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp; cd $ROOT
O=gpurun_out/b1; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/kt -o kt -- python bench.py --batch 1 --graphs --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2>/dev/null
python profiles/kernel_stats.py $O/kt/kt_results.db > $O/kernel_stats_b1.txt
rm -rf $O/kt
