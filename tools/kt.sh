#!/bin/bash
# usage (on the GPU box, from the repo root): tools/kt.sh <out_name> <command...>
# rocprofv3 --kernel-trace of the command; the per-kernel / per-grid summary goes to gpurun_out/<out_name>.txt
set -u
NAME=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
D=gpurun_out/kt_$NAME
rm -rf "$D"
rocprofv3 --kernel-trace -d "$D" -o kt -- "$@" > "$D.log" 2>&1
python profiles/kernel_stats.py "$D/kt_results.db" > "gpurun_out/$NAME.txt"
rm -rf "$D"
