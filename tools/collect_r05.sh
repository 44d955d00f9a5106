#!/bin/bash
# Round-5 measurement artefacts on the GPU box (run through gpurun from the repo root); outputs in gpurun_out/r05/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
O=gpurun_out/r05
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt_bench -o kt -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python profiles/kernel_stats.py $O/kt_bench/kt_results.db > $O/kernel_stats_bench.txt
cp $O/kt_bench/kt_kernel_stats.csv $O/rocprofv3_stats_bench.csv 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=$O/pmc_$(echo $c | cut -d' ' -f1)
  MCQUIC_AMD_BRANCH_STREAMS=0 rocprofv3 --pmc $c --kernel-trace -d $d -o pmc -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
done
python profiles/pmc_stats.py $O/pmc_FETCH_SIZE/pmc_results.db $O/pmc_WRITE_SIZE/pmc_results.db $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES/pmc_results.db > $O/pmc_by_kernel.txt 2>&1
cp /tmp/pmc_rows.json $O/pmc_rows.json 2>/dev/null
python profiles/make_pmc_json.py $O/pmc_rows.json > $O/pmc.json 2>$O/pmc_json.err
# the bench line reads the newest profiles/rNN_pmc.json for roofline.traffic: hand it THIS collection's passes (same kernel sources)
[ -s $O/pmc.json ] && cp $O/pmc.json profiles/r05_pmc.json
python bench.py > $O/bench.json 2> $O/bench.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_torchrun.json 2> $O/bench_torchrun.err
MCQUIC_AMD_BRANCH_STREAMS=0 rocprofv3 --kernel-trace -d $O/kt_bench_ss -o kt -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python profiles/kernel_stats.py $O/kt_bench_ss/kt_results.db > $O/kernel_stats_bench_single_stream.txt
python tools/bench_train.py --steps 10 2>/dev/null | tail -1 > $O/bench_train.json
python tools/bench_train.py --steps 20 --graph 2>/dev/null | tail -1 >> $O/bench_train.json
python tools/bench_train.py --steps 20 --graph --optimizer-step 2>/dev/null | tail -1 >> $O/bench_train.json
python tools/bench_train.py --steps 20 --graphed 2>/dev/null | tail -1 >> $O/bench_train.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 tools/bench_train.py --gpus 1 --steps 20 --graphed 2>/dev/null | tail -1 >> $O/bench_train.json
MCQUIC_AMD_BRANCH_STREAMS=0 rocprofv3 --kernel-trace -d $O/kt_train -o kt -- python tools/bench_train.py --graph --steps 6 > /dev/null 2>&1
python profiles/kernel_stats.py $O/kt_train/kt_results.db > $O/kernel_stats_train_graph.txt
python tools/train_conv_census.py --top 200 > $O/train_conv_census.txt 2>/dev/null
python tools/train_conv_census.py --eval --batch 32 --height 768 --width 512 --top 200 > $O/infer_conv_census.txt 2>/dev/null
python tools/bench_speed_protocol.py 2>/dev/null | tail -1 > $O/speed_protocol.txt
python tools/bench_metrics.py > $O/bench_metrics.txt 2>/dev/null
python tools/bench_vq.py > $O/bench_vq.json 2>/dev/null
python tools/bench_neon.py --train-batches 4,8 > $O/bench_neon.txt 2>&1
python tools/microbench_conv.py --neon --flags res > $O/neon_tile_sweep.txt 2>&1
python tools/probes/dump_step_sequence.py $O/kt_train/kt_results.db > $O/train_step_sequence.txt 2>&1
python tools/autograd_fanin.py > $O/autograd_fanin.txt 2>&1
python bench.py --batch 1 --graphs --steps 50 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_batch1_graphs.json
python bench.py --winograd 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_winograd2d.json
(time python -m pytest tests -q -m gpu) > $O/gputest.log 2>&1; echo "gputest rc=$?" >> $O/summary.txt
cp gpurun_out/parity_measurements.json $O/parity_measurements.json 2>/dev/null
rm -rf $O/kt_bench $O/kt_train $O/kt_bench_ss $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES
ls -la $O
