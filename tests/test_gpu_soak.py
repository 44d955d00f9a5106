"""tools/soak_determinism.py at a twentieth of its length: the same inputs through encode / decode (eager, ragged, hipGraph replays)
and through the captured training graph, every result bit-identical to the first.  The full soak (7150 runs, 95 s:
profiles/r04_soak_determinism.json) found no mismatch."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_repeated_runs_are_bit_identical(dev):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_determinism.py"), "--scale", "0.05"], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["all_identical"] and d["train_graph_b8_256"]["finite"], d
    assert d["encode_decode_b1_graphs"]["iterations"] == 150 and d["train_graph_b8_256"]["iterations"] == 50
