# round 5, headline residue (b), second form: park only the first 48 / 32 channel pairs (three workgroups per CU stay resident)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; rm -rf $O; mkdir -p $O
for v in x48 x32; do
  export MCQUIC_AMD_LIB=$PWD/mcquic_amd/variants/libmcquic_$v.so
  timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "gdn or GDN" > $O/tests_$v.log 2>&1; echo "$v tests rc=$?" >> $O/summary.txt
done
for i in 1 2; do for v in xoff main x48 x32; do
  if [ $v = main ]; then unset MCQUIC_AMD_LIB; else export MCQUIC_AMD_LIB=$PWD/mcquic_amd/variants/libmcquic_$v.so; fi
  echo "== $v" >> $O/micro.txt
  timeout 300 python tools/microbench_conv.py --k1 --flags gdn --tiles 0,0x41 2>/dev/null | grep -v "^lib" >> $O/micro.txt
done; done
for i in 1 2; do for v in xoff x48 x32; do
  export MCQUIC_AMD_LIB=$PWD/mcquic_amd/variants/libmcquic_$v.so
  timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])" >> $O/ab.txt
done; done
for v in x48 x32; do tail -2 $O/tests_$v.log; done; cat $O/summary.txt $O/micro.txt $O/ab.txt
