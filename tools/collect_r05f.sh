cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_step_ops.py tests/test_neon.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt
python tools/bench_neon.py --train-batches 4,8 > $O/bench_neon.txt 2>&1
tools/kt.sh r05_neon_train_dense python tools/prof_neon.py --train --dense
tail -4 $O/tests.log; cat $O/summary.txt; tail -4 $O/bench_neon.txt | cut -c1-900; head -14 gpurun_out/r05_neon_train_dense.txt | cut -c1-150
