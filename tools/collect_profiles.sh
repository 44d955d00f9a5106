#!/bin/bash
# Collect the round's measurement artefacts on the GPU box (run through gpurun from the repo root):
#   bench line, rocprofv3 kernel-trace summaries (branch streams on / off), PMC passes (separate runs), VQ / training benches.
# Outputs land in gpurun_out/final/; the summaries worth keeping are copied into profiles/ by hand.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
rm -rf $O && mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_torchrun.json 2> $O/bench_torchrun.err
rocprofv3 --kernel-trace -d $O/kt_branch -o kt -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python profiles/kernel_stats.py $O/kt_branch/kt_results.db > $O/kernel_stats_branch_streams.txt
MCQUIC_AMD_BRANCH_STREAMS=0 rocprofv3 --kernel-trace -d $O/kt_single -o kt -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python profiles/kernel_stats.py $O/kt_single/kt_results.db > $O/kernel_stats_single_stream.txt
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=$O/pmc_$(echo $c | cut -d' ' -f1)
  MCQUIC_AMD_BRANCH_STREAMS=0 rocprofv3 --pmc $c --kernel-trace -d $d -o pmc -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python profiles/pmc_stats.py $O/pmc_FETCH_SIZE/pmc_results.db $O/pmc_WRITE_SIZE/pmc_results.db $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES/pmc_results.db > $O/pmc_by_kernel.txt 2>&1
cp /tmp/pmc_rows.json $O/pmc_rows.json 2>/dev/null
python tools/bench_vq.py > $O/bench_vq.json 2>/dev/null
rocprofv3 --kernel-trace -d $O/kt_vq -o kt -- python tools/bench_vq.py > /dev/null 2>&1
python profiles/kernel_stats.py $O/kt_vq/kt_results.db > $O/kernel_stats_vq.txt
python tools/bench_train.py 2>/dev/null | tail -1 > $O/bench_train.json
rocprofv3 --kernel-trace -d $O/kt_train -o kt -- python tools/bench_train.py > /dev/null 2>&1
python profiles/kernel_stats.py $O/kt_train/kt_results.db > $O/kernel_stats_train.txt
python tools/bench_metrics.py > $O/bench_metrics.txt 2>/dev/null
rm -rf $O/kt_* $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES
ls -la $O
tail -1 $O/bench.json | cut -c1-400
