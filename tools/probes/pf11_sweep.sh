# kernel durations (rocprofv3) of the tiny-map launches for ring-depth variants of the 32x32 tile; host-side timing cannot resolve them
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pf11
for L in default pf11_18_36 pf11_36_36; do
  if [ $L = default ]; then unset MCQUIC_AMD_LIB; else export MCQUIC_AMD_LIB=$PWD/mcquic_amd/variants/$L.so; fi
  for np in 2 4; do
    rm -rf /tmp/kt; rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python tools/microbench_conv.py --train --nprob $np --iters 30 > /dev/null 2>&1
    echo "== $L nprob $np"; python profiles/kernel_stats.py /tmp/kt/kt_results.db | grep "conv_mfma_kernel<1, 1" | cut -c1-150
  done
done
