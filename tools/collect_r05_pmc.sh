#!/bin/bash
# The counter passes + the bench line + the GPU tests only (after a change that leaves the kernels' code as it was but moves csrc_sha:
# roofline.traffic must come from passes over the sources the bench line runs on).  Outputs in gpurun_out/r05pmc/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
O=gpurun_out/r05pmc
rm -rf $O && mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=$O/pmc_$(echo $c | cut -d' ' -f1)
  MCQUIC_AMD_BRANCH_STREAMS=0 rocprofv3 --pmc $c --kernel-trace -d $d -o pmc -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
done
python profiles/pmc_stats.py $O/pmc_FETCH_SIZE/pmc_results.db $O/pmc_WRITE_SIZE/pmc_results.db $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES/pmc_results.db > $O/pmc_by_kernel.txt 2>&1
cp /tmp/pmc_rows.json $O/pmc_rows.json 2>/dev/null
python profiles/make_pmc_json.py $O/pmc_rows.json > $O/pmc.json 2>$O/pmc_json.err
[ -s $O/pmc.json ] && cp $O/pmc.json profiles/r05_pmc.json
python bench.py > $O/bench.json 2> $O/bench.err
(time python -m pytest tests -q -m gpu) > $O/gputest.log 2>&1; echo "gputest rc=$?" >> $O/summary.txt
cp gpurun_out/parity_measurements.json $O/parity_measurements.json 2>/dev/null
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES
tail -3 $O/gputest.log; cat $O/summary.txt; cut -c1-300 $O/bench.json
