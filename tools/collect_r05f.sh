cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_step_ops.py tests/test_neon.py tests/test_gpu_backward.py tests/test_gpu_golden.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt
python tools/bench_neon.py --train-batches 4 > $O/bench_neon.txt 2>&1
MCQUIC_AMD_NORM_NODES=0 python tools/bench_neon.py --train-batches 4 > $O/bench_neon_opbyop.txt 2>&1
tail -8 $O/tests.log; cat $O/summary.txt; tail -2 $O/bench_neon.txt; tail -2 $O/bench_neon_opbyop.txt | head -1
