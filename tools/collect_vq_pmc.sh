ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp; cd $ROOT
O=gpurun_out/vqpmc; rm -rf $O; mkdir -p $O
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  d=$O/pmc_$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $d -o pmc -- python tools/bench_vq.py > $O/bench_vq.json 2>/dev/null
done
python profiles/pmc_stats.py $O/pmc_FETCH_SIZE/pmc_results.db $O/pmc_WRITE_SIZE/pmc_results.db $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES/pmc_results.db $(ls $O/pmc_SQ_INSTS_VALU/pmc_results.db 2>/dev/null) > $O/pmc_by_kernel_vq.txt 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES $O/pmc_SQ_INSTS_VALU
cat $O/pmc_by_kernel_vq.txt | head -20; tail -1 $O/bench_vq.json | cut -c1-400
