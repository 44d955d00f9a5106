cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_graphed_step.py tests/test_gpu_two_ranks.py tests/test_gpu_train_rehearsal.py tests/test_host_abi.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt
for i in 1 2; do
python tools/bench_train.py --graphed --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graphed sgd', d['ms_per_step'])" >> $O/graphed.txt
done
python - > $O/adam.txt 2>&1 <<'PY'
import sys, os, torch
sys.path.insert(0, os.getcwd())
os.environ["MCQUIC_AMD_BRANCH_STREAMS"] = "0"
import bench
from mcquic_amd import Compressor, optim, parallel
dev = torch.device("cuda:0")
torch.manual_seed(3407)
tm = Compressor(128, 2, [8192, 2048, 512]).to(dev).train()
xt = (torch.rand((8, 3, 256, 256), generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
g = parallel.GraphedTrainStep(tm, optim.Adam(tm.parameters(), lr=torch.tensor(1e-6, device=dev)), xt, max_grad_norm=4.0)
print("graphed adam+clip", round(bench._timed(lambda: g(xt), 20, warmup=3), 3))
PY
tail -5 $O/tests.log; cat $O/summary.txt $O/graphed.txt $O/adam.txt
