#!/bin/bash
# PMC passes over the captured training step (tools/bench_train.py --graph): HBM bytes and MFMA-pipe busy cycles per kernel; output
# gpurun_out/pmc_train_by_kernel.txt (copy to profiles/rNN_pmc_train_by_kernel.txt).  Counters in their own runs, kernel trace only.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
O=gpurun_out/pmc_train
rm -rf $O && mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=$O/$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace -d $d -o pmc -- python tools/bench_train.py --graph --steps 2 --warmup 1 > /dev/null 2>&1
done
PMC_KERNELS=conv_mfma,conv_t16,conv_wgrad,wgrad_rows,vq_logits,vq_gumbel,vq_softmax,vq_dx,vq_dc python profiles/pmc_stats.py $O/FETCH_SIZE/pmc_results.db $O/WRITE_SIZE/pmc_results.db $O/SQ_VALU_MFMA_BUSY_CYCLES/pmc_results.db > gpurun_out/pmc_train_by_kernel.txt 2>&1
rm -rf $O
head -50 gpurun_out/pmc_train_by_kernel.txt
