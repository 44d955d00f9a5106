cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
python tools/microbench_conv.py --batch1 --flags res > $O/b1_tiles_res.txt 2>&1
python tools/microbench_conv.py --batch1 --flags silu_out > $O/b1_tiles_silu.txt 2>&1
python tools/microbench_conv.py --batch1 --flags res --nprob 2 > $O/b1_tiles_res_np2.txt 2>&1
python tools/probe_weight_warmth.py > $O/weight_warmth.txt 2>&1
cat $O/b1_tiles_res.txt $O/b1_tiles_res_np2.txt $O/weight_warmth.txt
