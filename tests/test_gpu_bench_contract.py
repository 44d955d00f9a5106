"""GPU: the bench.py output contract (one JSON line, required keys, roofline / cpu_baseline objects) on a tiny run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2",
                          *extra], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert lines, "no output"
    return json.loads(lines[-1])                 # the JSON line is the last thing printed


def test_bench_line_has_the_contract_fields(dev):
    d = _run()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 2 * 1000.0 / d["ms_per_step"]) < 1e-2 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0 < r["frac"] <= 1.0 and 0 < r["whole_step_frac"] <= 1.0
    assert r["traffic_stale"] in (True, False, None)
    # the round's other results ride in the same line, each with a fraction of the peak that is a true
    # utilisation (real issued work / time / peak), never above 1
    sec = d["secondary"]
    for key in ("winograd2d", "train_step", "vq_config4", "batch1", "host_buffers", "speed_protocol"):
        assert key in sec and "error" not in sec[key], (key, sec.get(key))
    assert 0 < sec["winograd2d"]["mfma_work_frac"] <= 1.0 and sec["winograd2d"]["images_s"] > 0
    assert sec["winograd2d"]["mfma_gflop_per_step"] < sec["winograd2d"]["direct_form_gflop_per_step"]
    assert sec["winograd2d"]["parity"]["decode_max_abs_err"] <= 1e-4
    assert sec["train_step"]["graph"] is True and 0 < sec["train_step"]["frac_of_peak"] <= 1.0
    assert 0 < sec["vq_config4"]["frac_of_peak"] <= 1.0 and 0 < sec["batch1"]["frac_of_peak"] <= 1.0
    # the PCIe-inclusive rate is reported beside the headline, never as it (at this test's 2 images x 2 steps the two are not
    # comparable: the child process times warmer steps)
    assert sec["host_buffers"]["images_s"] > 0 and sec["host_buffers"]["h2d_mb_per_step"] > 0
    assert sec["speed_protocol"]["encode_mpps"] > 0 and sec["speed_protocol"]["decode_mpps"] > 0
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1
    assert d["parity"]["code_mismatches"] == 0


def test_opt_in_modes_report_honest_roofline_fields(dev):
    """--winograd: `frac` is the issued MFMA work over the peak (<= 1; the direct-form-equivalent rate sits beside it);
    --graphs: nothing to bracket, so the per-kernel fields are null, not 0.0."""
    w = _run("--no-cpu-baseline", "--winograd", "2")              # (opt-in modes never carry `secondary`)
    assert "secondary" not in w
    r = w["roofline"]
    assert 0 < r["frac"] <= 1.0 and r["equivalent_direct"] > r["achieved"] and r["mfma_gflop_per_step"] < r["algorithmic_gflop_per_step"]
    g = _run("--no-cpu-baseline", "--graphs")
    r = g["roofline"]
    assert r["achieved"] is None and r["frac"] is None and r["launches_per_step"] is None and r["avg_launch_ms"] is None
    assert 0 < r["whole_step_frac"] <= 1.0 and g["value"] > 0


def test_bench_under_a_process_group(dev):
    """RANK / WORLD_SIZE in the environment (what torch.distributed.run sets): RCCL path at world size 1."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2",
                          "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "cpu_baseline" not in d
    assert d["rccl_world"] == 1                              # an RCCL all-reduce actually ran over the process group
    # every rank's own time before the closing barrier rides along; the contract's figure (barrier to barrier, MAX) bounds it
    assert len(d["rank_ms_per_step"]) == 1 and 0 < d["rank_ms_per_step"][0] <= d["ms_per_step"] * 1.001


def test_bench_without_a_launcher_takes_no_process_group(dev):
    d = _run("--no-cpu-baseline", "--no-secondary")
    assert d["n_gpus"] == 1 and d["rccl_world"] is None and d["rank_ms_per_step"] is None and "secondary" not in d


def test_bench_gpus_must_match_the_launcher(dev):
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29579")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, cwd=ROOT, timeout=300, env=env)
    assert out.returncode == 2


def test_import_turns_packet_recorded_graph_launches_off():
    """mcquic_amd/__init__.py: ROCm 7.2 replays memset nodes of captured graphs wrongly on its packet-recorded launch path; the
    import switches that path off unless the caller has chosen (tests/test_gpu_graph_replay.py holds the behaviour itself)."""
    code = "import os; os.environ.pop('DEBUG_CLR_GRAPH_PACKET_CAPTURE', None); import mcquic_amd; print(os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'])"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "0", out.stderr[-500:]
    kept = subprocess.run([sys.executable, "-c", code.replace("os.environ.pop('DEBUG_CLR_GRAPH_PACKET_CAPTURE', None)",
                                                                "os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] = '1'")],
                          capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert kept.returncode == 0 and kept.stdout.strip() == "1", kept.stderr[-500:]


def test_import_sets_hw_queue_default():
    out = subprocess.run([sys.executable, "-c", "import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); import mcquic_amd; "
                          "print(os.environ['GPU_MAX_HW_QUEUES'])"], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "8", out.stderr[-500:]


def test_training_step_under_ddp_matches_the_plain_step(dev):
    """BASELINE configs[4] wraps the model in torch DDP over RCCL.  One rank is all this box has: the step is run under
    `torch.distributed.run --nproc-per-node 1` (DDP buckets, gradient hooks and an RCCL all-reduce per bucket are live) and
    must give the loss and gradient norm of the plain step."""
    script = os.path.join(ROOT, "tools", "bench_train.py")
    common = ["--steps", "2", "--warmup", "1", "--batch", "2", "--crop", "128"]
    plain = subprocess.run([sys.executable, script, *common], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert plain.returncode == 0, plain.stderr[-2000:]
    ddp = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                          "--master-port", "29581", script, "--gpus", "1", *common], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert ddp.returncode == 0, ddp.stderr[-2000:]
    a = json.loads([ln for ln in plain.stdout.splitlines() if ln.startswith("{")][-1])
    b = json.loads([ln for ln in ddp.stdout.splitlines() if ln.startswith("{")][-1])
    assert b["ddp"] is True and a["ddp"] is False
    assert abs(a["loss"] - b["loss"]) <= 1e-6 * abs(a["loss"]) and abs(a["grad_norm"] - b["grad_norm"]) <= 1e-5 * a["grad_norm"]


def test_bench_gpus_2_really_launches_two_ranks(dev):
    """`python bench.py --gpus 2` with no launcher re-executes itself under torch.distributed.run with two ranks (VERDICT r1:
    the flag used to be parsed and ignored).  This box has one GPU, so the ranks must refuse -- loudly, naming the cause --
    instead of quietly measuring one GPU; on a node with two GPUs the same command is the N = 2 benchmark."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU node: the command would run the real N = 2 benchmark")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode != 0
    assert "--gpus 2 but this node has 1 HIP devices" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]      # no benchmark line from a refused run
