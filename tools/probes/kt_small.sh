cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/kt_small
for lib in default mcquic_amd/variants/pf36.so; do
  tag=$(basename $lib .so)
  for np in 1 4; do
    d=/tmp/kt_${tag}_$np; rm -rf $d
    L=$lib; [ $lib = default ] && L=""
    MCQUIC_AMD_LIB=$L rocprofv3 --kernel-trace -d $d -o kt -- python tools/microbench_conv.py --train --small --nprob $np --tiles 0 > /dev/null 2>&1
    python profiles/kernel_stats.py $d/kt_results.db > gpurun_out/kt_small/${tag}_np$np.txt 2>&1
  done
done
grep -h "conv_mfma" gpurun_out/kt_small/*.txt | head -5
