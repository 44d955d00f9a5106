#!/bin/bash
# Round-2 measurement artefacts on the GPU box (run through gpurun from the repo root); outputs in gpurun_out/r02/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
O=gpurun_out/r02
rm -rf $O && mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_torchrun.json 2> $O/bench_torchrun.err
rocprofv3 --kernel-trace -d $O/kt_bench -o kt -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python profiles/kernel_stats.py $O/kt_bench/kt_results.db > $O/kernel_stats_bench.txt
python tools/bench_train.py --steps 10 2>/dev/null | tail -1 > $O/bench_train.json
python tools/bench_train.py --steps 10 --graph 2>/dev/null | tail -1 >> $O/bench_train.json
python tools/bench_train.py --steps 10 --optimizer-step 2>/dev/null | tail -1 >> $O/bench_train.json
MCQUIC_AMD_BRANCH_STREAMS=0 rocprofv3 --kernel-trace -d $O/kt_train -o kt -- python tools/bench_train.py > /dev/null 2>&1
python profiles/kernel_stats.py $O/kt_train/kt_results.db > $O/kernel_stats_train_single_stream.txt
python tools/bench_speed_protocol.py 2>/dev/null | tail -1 > $O/speed_protocol.txt
python tools/bench_vq.py > $O/bench_vq.json 2>/dev/null
rm -rf $O/kt_bench $O/kt_train
ls -la $O
