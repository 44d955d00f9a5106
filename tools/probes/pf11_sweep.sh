# HISTORICAL (round 3): kernel durations (rocprofv3) of the tiny-map launches for ring-depth variants of the 32x32 conv tile.
# The -DMCQ_PF11A / -DMCQ_PF11B build switches it used were removed with the negative result (10.2 -> 10.8 us, training step
# 24.1 -> 24.2 ms); to repeat it, change the two ring depths in conv_mfma.hip's launch_tile<1, 1, 9, MCQ_PFB, 16> line.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for np in 2 4; do
  rm -rf /tmp/kt; rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python tools/microbench_conv.py --train --nprob $np --iters 30 > /dev/null 2>&1
  echo "== nprob $np"; python profiles/kernel_stats.py /tmp/kt/kt_results.db | grep "conv_mfma_kernel<1, 1" | cut -c1-150
done
