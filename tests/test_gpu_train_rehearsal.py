"""A training LOOP through parallel.GraphedTrainStep (tools/train_rehearsal.py): mcquic_amd.optim.Adam with a scheduled learning
rate in a device tensor, gradient clipping by global norm (the reference's step: mcquic/train/trainer.py:263-296), `loss.item()`
every step, and every 50 steps what the reference's hooks do between steps -- finiteness of every parameter, a validation pass
through the eager encode / decode, codebook re-assignment (mcquic/train/hooks.py:100-121) -- on synthetic images a model can learn.
The full-size run (Compressor(128, 2, [8192, 2048, 512]), 2000 steps: loss 0.182 -> 0.074, PSNR 13.3 -> 16.9 dB, 25.1 ms per step)
is profiles/r04_train_rehearsal.json; here a small model for 200 steps."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_training_loop_learns_and_stays_finite(dev):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "train_rehearsal.py"), "--channel", "32", "--ks", "64,32,16", "--crop", "64", "--batch", "4",
           "--steps", "200", "--every", "50", "--warmup", "20", "--lr", "1e-3", "--own-adam"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["post_captured"], "the one-launch Adam + clipping should have been captured with the update"
    assert d["memset_nodes_ok"] and d["non_finite"] == 0
    assert d["loss_last"] < 0.9 * d["loss_first"], (d["loss_first"], d["loss_last"])          # measured 0.181 -> 0.148
    assert d["psnr_last"] > d["psnr_first"] + 0.5, d["psnr"]                                   # measured 13.30 -> 14.37 dB
    assert len(d["reassigned"]) == 1 and all(g == g and g > 0 for _, g in d["grad_norm"])
