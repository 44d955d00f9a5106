cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/neon_trace
for d in 1 0; do
  rm -rf /tmp/ktn_$d
  rocprofv3 --kernel-trace -d /tmp/ktn_$d -o kt -- python tools/probes/neon_infer_trace.py $d > /dev/null 2>&1
  python profiles/kernel_stats.py /tmp/ktn_$d/kt_results.db > gpurun_out/neon_trace/dense$d.txt 2>&1
done
head -20 gpurun_out/neon_trace/dense1.txt | cut -c1-150
