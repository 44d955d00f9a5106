# round 5, headline residue (b): GDN / IGDN 1x1 launches keep the streamed x in LDS instead of reading it again
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05j; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_golden.py tests/test_gpu_model.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt
for v in main xoff xt1 xt0; do
  if [ $v = main ]; then unset MCQUIC_AMD_LIB; else export MCQUIC_AMD_LIB=$PWD/mcquic_amd/variants/libmcquic_$v.so; fi
  echo "== $v" >> $O/micro.txt
  for fl in gdn igdn; do timeout 300 python tools/microbench_conv.py --k1 --flags $fl >> $O/micro.txt 2>&1; done
done
for i in 1 2; do for v in main xoff xt1 xt0; do
  if [ $v = main ]; then unset MCQUIC_AMD_LIB; else export MCQUIC_AMD_LIB=$PWD/mcquic_amd/variants/libmcquic_$v.so; fi
  timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])" >> $O/ab.txt
done; done
tail -3 $O/tests.log; cat $O/summary.txt $O/micro.txt $O/ab.txt
