# GroupNorm in one launch per direction against two (MCQUIC_AMD_GN_FUSED=0): tests first, then the Neon figures alternating
timeout 600 python -m pytest tests/test_gpu_step_ops.py -x -q -k "group_norm" 2>&1 | tail -4
for r in 1 2; do for f in 0 1; do
  MCQUIC_AMD_GN_FUSED=$f timeout 400 python - <<'PY' 2>&1 | tail -1
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
import bench
r = bench.neon_figures(torch.device("cuda:0"), dense=True, train_batch=4, infer_batch=8, side=512)
print("GN_FUSED=" + os.environ["MCQUIC_AMD_GN_FUSED"], {k: r[k] for k in ("images_s", "ms_per_step", "train_step_graph_ms", "train_step_ms", "train_loss")})
PY
done; done
