"""Replays of the captured training step with eager work in between, under BOTH hipGraph launch paths of the runtime.

ROCm 7.2 records the launch packets of a graph when it is instantiated (DEBUG_CLR_GRAPH_PACKET_CAPTURE, on by default); the
packet of a memset NODE points at blit arguments that later eager blit work (small device-to-host copies, fills) reuses, so the
node zeroes the wrong thing on the next replay.  Round 4 met it twice: the soft assignment's input gradient (a hipMemsetAsync in
front of an atomic accumulation) turned non-finite after `loss.item()`-style work between replays, and ATen's two-stage reduction
behind F.mse_loss returned the FIRST replay's loss forever (docs/experiments.md section 9.9).  The captured step now holds no
memset node -- checked here by forcing the packet path ON in a child process -- and mcquic_amd/__init__.py turns the packet path
off when it is imported before the HIP runtime starts."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(packet_capture: str):
    env = dict(os.environ, DEBUG_CLR_GRAPH_PACKET_CAPTURE=packet_capture)
    res = subprocess.run([sys.executable, os.path.join(HERE, "_graph_replay_worker.py")], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    return json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.fixture(scope="module")
def both(dev):
    return _run("1"), _run("0")


def test_step_survives_eager_work_on_both_launch_paths(both):
    for r in both:
        assert r["first_replay_finite"]
        assert r["non_finite_tensors_per_step"] == [0] * 6, r
        assert r["replay_vs_first_worst_relative"] <= 1e-4, r          # (measured ~1e-6: two atomic addends in either order)
        assert len(set(r["losses"])) == 6, f"a replay returned an earlier replay's loss: {r['losses']}"
        assert r["inference_graph_replays_equal_eager"] == [True] * 8, r
    on, off = both
    for a, b in zip(on["losses"], off["losses"]):
        assert abs(a - b) <= 2e-6 * abs(b), (on["losses"], off["losses"])


def test_memset_node_check_reports_the_launch_path(both):
    """With the packet path off the check must pass (that is what importing mcquic_amd first buys); with it on, ROCm 7.2 fails
    it -- recorded, not asserted: a runtime that fixes the defect is welcome."""
    on, off = both
    assert off["memset_nodes_ok"] is True
    from _record import record
    record("memset_nodes_replay_correctly", packet_capture_on=on["memset_nodes_ok"], packet_capture_off=off["memset_nodes_ok"],
           rocm=str(torch.version.hip))


def test_mse_matches_torch(dev):
    from mcquic_amd import ops
    from mcquic_amd.autograd import mse_loss
    for n in (1, 255, 4096, 8 * 3 * 256 * 256, 3 * 1000 * 777):
        g = torch.Generator().manual_seed(n)
        a = torch.randn(n, generator=g).to(dev).requires_grad_(True)
        b = torch.randn(n, generator=g).to(dev).requires_grad_(True)
        a2, b2 = a.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
        want = torch.nn.functional.mse_loss(a2, b2)
        got = mse_loss(a, b)
        assert got.shape == () and got.dtype == torch.float32
        assert abs(float(got.detach()) - float(want.detach())) <= 2e-7 * float(want.detach()) + 1e-12
        (got * 3.0).backward()
        (want * 3.0).backward()
        assert float((a.grad.double() - a2.grad).abs().max()) <= 1e-6 * float(a2.grad.abs().max())
        assert torch.equal(b.grad, -a.grad)
        assert torch.equal(ops.mse(a.detach(), b.detach()), got.detach())           # a fixed summation order: bit-reproducible
