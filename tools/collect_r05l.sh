# round 5, headline residue (b), third form: the GDN / IGDN epilogue's 1/sqrt and sqrt from v_rsq_f32 + one Newton step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l; rm -rf $O; mkdir -p $O
export MCQUIC_AMD_LIB=$PWD/mcquic_amd/variants/libmcquic_frsq.so
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_golden.py tests/test_gpu_model.py -q -m gpu > $O/tests_frsq.log 2>&1; echo "frsq tests rc=$?" >> $O/summary.txt
for i in 1 2; do for v in main frsq; do
  if [ $v = main ]; then unset MCQUIC_AMD_LIB; else export MCQUIC_AMD_LIB=$PWD/mcquic_amd/variants/libmcquic_$v.so; fi
  echo "== $v" >> $O/micro.txt
  timeout 300 python tools/microbench_conv.py --k1 --flags gdn --tiles 0,0x41 2>/dev/null | grep -v "^lib" >> $O/micro.txt
done; done
for i in 1 2 3; do for v in main frsq; do
  if [ $v = main ]; then unset MCQUIC_AMD_LIB; else export MCQUIC_AMD_LIB=$PWD/mcquic_amd/variants/libmcquic_$v.so; fi
  timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])" >> $O/ab.txt
done; done
tail -4 $O/tests_frsq.log; cat $O/summary.txt $O/micro.txt $O/ab.txt
