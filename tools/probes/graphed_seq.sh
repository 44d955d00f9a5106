cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/kt -o kt -- python tools/bench_train.py --graphed --steps 6 > $O/kt.log 2>&1
python tools/probes/dump_step_sequence.py $O/kt/kt_results.db > $O/graphed_sequence.txt 2>&1
rm -rf $O/kt
head -2 $O/graphed_sequence.txt
python tools/step_buckets.py $O/graphed_sequence.txt
grep -n "pack_conv\|reparam\|sgd\|adam\|Cat\|at::native\|rocclr\|freq_ema\|multi_tensor" $O/graphed_sequence.txt | awk '{print $3, $4, $5, $6}' | sort -k3 | uniq -c -f2 | head -40
