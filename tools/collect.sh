#!/bin/bash
# ONE entry point for the measurement artefacts of a round, run on the GPU box through gpurun from the repo root:
#
#     gpurun --timeout 2400 -- 'bash tools/collect.sh <what> [round]'          (round = rNN prefix of the outputs, default r06)
#
#   final    everything a round commits under profiles/: bench line, torchrun line, rocprofv3 kernel-trace summaries of the same
#            command (branch streams on / single stream), the three PMC passes (separate runs, --kernel-trace only) -> rNN_pmc.json
#            (stamped with the kernel sources' hash: bench.py reads it for roofline.traffic), conv censuses, training step
#            (eager / one hipGraph / with the optimizer / GraphedTrainStep; trace + launch sequence + buckets), VQ, Neon, batch 1,
#            speed protocol, metrics, the GPU test suite with its durations
#   pmc      only the counter passes + the bench line (after a change that moves the kernel sources' hash)
#   bench    only the bench line (+ the torchrun line)
#   train    training step figures + trace + buckets
#   tests    the GPU test suite with durations
#   vqpmc    counter passes over tools/bench_vq.py (vq_assign launches: fabric reads / writes, MFMA-busy, VALU / MFMA instruction counts)
#   abtrain VAR A B [runs]   the same on the captured training step (tools/bench_train.py --graph)
#   ab VAR A B [runs]   alternating `VAR=A` / `VAR=B` headline runs on this box (bench.py --steps 8 --no-secondary): in-step A/B of a switch
# Outputs: gpurun_out/<round>_<what>/ (merged back by gpurun; what the script copies into profiles/ on the BOX only serves the same call --
# bench.py reads profiles/rNN_pmc.json -- and is copied from gpurun_out/ into profiles/ again by hand after the call)
# (kernel traces and counter databases are deleted: too large to travel).  Earlier rounds' one-off scripts: git history (tools/collect_r0*.sh).
set -u
WHAT=${1:-final}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"

kt() {          # kt <out.txt> <env...> -- <command...>: rocprofv3 --kernel-trace of a command, summarised per kernel / launch grid
  local out=$1; shift
  local d=$O/kt_$$; rm -rf $d
  env "$@" rocprofv3 --kernel-trace -d $d -o kt -- "${CMD[@]}" > /dev/null 2>&1
  python profiles/kernel_stats.py $d/kt_results.db > $out 2>&1
  KT_DB=$d/kt_results.db
}

pmc_passes() {  # FETCH_SIZE / WRITE_SIZE / MFMA-busy, one pass each (gpurun refuses --pmc together with other trace domains)
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    d=$O/pmc_$(echo $c | cut -d' ' -f1)
    MCQUIC_AMD_BRANCH_STREAMS=0 rocprofv3 --pmc $c --kernel-trace -d $d -o pmc -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  done
  python profiles/pmc_stats.py $O/pmc_FETCH_SIZE/pmc_results.db $O/pmc_WRITE_SIZE/pmc_results.db $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES/pmc_results.db > $O/pmc_by_kernel.txt 2>&1
  cp /tmp/pmc_rows.json $O/pmc_rows.json 2>/dev/null
  python profiles/make_pmc_json.py $O/pmc_rows.json > $O/pmc.json 2>$O/pmc_json.err
  [ -s $O/pmc.json ] && cp $O/pmc.json profiles/${R}_pmc.json && cp $O/pmc_by_kernel.txt profiles/${R}_pmc_by_kernel.txt
  rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES
}

bench_lines() {
  python bench.py > $O/bench.json 2> $O/bench.err
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_torchrun.json 2> $O/bench_torchrun.err
  [ -s $O/bench.json ] && tail -1 $O/bench.json > profiles/${R}_bench.json
  [ -s $O/bench_torchrun.json ] && tail -1 $O/bench_torchrun.json > profiles/${R}_bench_torchrun.json
}

train_figures() {
  python tools/bench_train.py --steps 10 2>/dev/null | tail -1 > $O/bench_train.json
  python tools/bench_train.py --steps 20 --graph 2>/dev/null | tail -1 >> $O/bench_train.json
  python tools/bench_train.py --steps 20 --graph --optimizer-step 2>/dev/null | tail -1 >> $O/bench_train.json
  python tools/bench_train.py --steps 20 --graphed 2>/dev/null | tail -1 >> $O/bench_train.json
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 tools/bench_train.py --gpus 1 --steps 20 --graphed 2>/dev/null | tail -1 >> $O/bench_train.json
  CMD=(python tools/bench_train.py --graph --steps 6); kt $O/kernel_stats_train_graph.txt MCQUIC_AMD_BRANCH_STREAMS=0
  python tools/probes/dump_step_sequence.py $KT_DB > $O/train_step_sequence.txt 2>&1
  python tools/step_buckets.py $O/train_step_sequence.txt > $O/train_step_buckets.txt 2>&1
  rm -rf $O/kt_$$
  python tools/train_conv_census.py --top 200 > $O/train_conv_census.txt 2>/dev/null
  for f in bench_train.json kernel_stats_train_graph.txt train_step_sequence.txt train_step_buckets.txt train_conv_census.txt; do cp $O/$f profiles/${R}_$f; done
}

gpu_tests() {
  (time python -m pytest tests -q -m gpu --durations=40) > $O/gputest.log 2>&1; echo "gputest rc=$?" >> $O/gputest.log
  cp gpurun_out/parity_measurements.json profiles/${R}_parity_measurements.json 2>/dev/null
  tail -60 $O/gputest.log > profiles/${R}_gputest_tail.txt
}

case "$WHAT" in
  final|pmc|bench|train|tests)
    R=${1:-r06}; O=gpurun_out/${R}_$WHAT; rm -rf $O; mkdir -p $O ;;
esac

case "$WHAT" in
  final)
    pmc_passes                      # (first: the bench line below then finds THIS collection's rNN_pmc.json)
    bench_lines
    CMD=(python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary)
    kt $O/kernel_stats_bench.txt MCQUIC_AMD_BRANCH_STREAMS=1; rm -rf $O/kt_$$
    kt $O/kernel_stats_bench_single_stream.txt MCQUIC_AMD_BRANCH_STREAMS=0; rm -rf $O/kt_$$
    python tools/train_conv_census.py --eval --batch 32 --height 768 --width 512 --top 200 > $O/infer_conv_census.txt 2>/dev/null
    train_figures
    python tools/bench_speed_protocol.py 2>/dev/null | tail -1 > $O/speed_protocol.txt
    MCQUIC_AMD_CODER_OVERLAP=0 python tools/bench_speed_protocol.py 2>/dev/null | tail -1 | sed 's/^/coder overlap off: /' >> $O/speed_protocol.txt
    python tools/bench_metrics.py > $O/bench_metrics.txt 2>/dev/null
    python tools/bench_vq.py > $O/bench_vq.json 2>/dev/null
    python tools/bench_post.py > $O/bench_post.txt 2>/dev/null; python tools/bench_post.py --batch1 >> $O/bench_post.txt 2>/dev/null
    python tools/bench_neon.py --train-batches 4,8 > $O/bench_neon.txt 2>&1
    python bench.py --batch 1 --graphs --steps 50 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_batch1_graphs.json
    CMD=(python bench.py --batch 1 --graphs --steps 20 --warmup 5 --no-cpu-baseline --no-secondary); kt $O/kernel_stats_batch1.txt MCQUIC_AMD_BRANCH_STREAMS=1; rm -rf $O/kt_$$
    for f in kernel_stats_bench.txt kernel_stats_bench_single_stream.txt infer_conv_census.txt speed_protocol.txt bench_metrics.txt bench_vq.json bench_post.txt bench_neon.txt bench_batch1_graphs.json kernel_stats_batch1.txt; do
      [ -s $O/$f ] && cp $O/$f profiles/${R}_$f
    done
    gpu_tests
    ls -la $O; tail -3 $O/gputest.log; cut -c1-400 $O/bench.json ;;
  pmc)   pmc_passes; bench_lines; cut -c1-400 $O/bench.json ;;
  bench) bench_lines; cut -c1-400 $O/bench.json ;;
  train) train_figures; cat $O/train_step_buckets.txt; cat $O/bench_train.json | cut -c1-300 ;;
  tests) gpu_tests; tail -50 $O/gputest.log ;;
  ab)
    VAR=$1; A=$2; B=$3; RUNS=${4:-2}; O=gpurun_out/ab_$VAR; mkdir -p $O
    for r in $(seq $RUNS); do for v in "$A" "$B"; do
      env $VAR=$v python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/ab.txt
    done; done ;;
  vqpmc)        # counter passes over tools/bench_vq.py: fabric reads / writes / MFMA-busy / instruction counts per vq_assign launch
    R=${1:-r06}; O=gpurun_out/${R}_vqpmc; rm -rf $O; mkdir -p $O
    for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA"; do
      d=$O/pmc_$(echo $c | cut -d' ' -f1)
      timeout 300 rocprofv3 --pmc $c --kernel-trace -d $d -o pmc -- python tools/bench_vq.py > $O/bench_vq.json 2>/dev/null
    done
    python profiles/pmc_stats.py $O/pmc_FETCH_SIZE/pmc_results.db $O/pmc_WRITE_SIZE/pmc_results.db $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES/pmc_results.db $(ls $O/pmc_SQ_INSTS_VALU/pmc_results.db 2>/dev/null) > $O/pmc_by_kernel_vq.txt 2>&1
    rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES $O/pmc_SQ_INSTS_VALU
    head -20 $O/pmc_by_kernel_vq.txt; tail -1 $O/bench_vq.json | cut -c1-600 ;;
  abtrain)      # the same for the captured training step (configs[4]): tools/bench_train.py --graph, alternating
    VAR=$1; A=$2; B=$3; RUNS=${4:-2}; O=gpurun_out/abtrain_$VAR; mkdir -p $O
    for r in $(seq $RUNS); do for v in "$A" "$B"; do
      env $VAR=$v python tools/bench_train.py --steps 20 --graph 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_step'], d['loss'], d['grad_norm'])" | tee -a $O/ab.txt
    done; done ;;
  *) echo "usage: tools/collect.sh final|pmc|bench|train|tests [rNN]  |  ab VAR A B [runs]"; exit 2 ;;
esac
