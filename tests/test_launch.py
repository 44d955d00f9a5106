"""CPU: `--gpus N` is the world size (mcquic_amd/launch.py) -- launcher argv, WORLD_SIZE mismatch, per-rank core plan."""
import os
import subprocess
import sys

import pytest

from mcquic_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_argv_is_one_rank_per_gpu():
    argv = launch.launcher_argv(8, "/x/bench.py", ["--gpus", "8", "--steps", "5"], port=29511, python="python3")
    assert argv == ["python3", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                    "--master-port", "29511", "/x/bench.py", "--gpus", "8", "--steps", "5"]
    with pytest.raises(ValueError):
        launch.launcher_argv(0, "/x/bench.py", [])


def test_ensure_world_reexecs_without_a_launcher():
    calls = []
    with pytest.raises(SystemExit):
        launch.ensure_world(4, "/x/bench.py", ["--gpus", "4"], env={"PATH": "/bin"}, _exec=lambda f, a, e: calls.append((f, a, e)))
    (f, argv, env), = calls
    assert f == argv[0] and "--nproc-per-node=4" in argv and argv[-3:] == ["/x/bench.py", "--gpus", "4"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_ensure_world_single_gpu_needs_no_launcher():
    assert launch.ensure_world(1, "/x/bench.py", [], env={}, _exec=None) == (0, 0, 1, False)


def test_ensure_world_inside_a_launcher():
    env = {"RANK": "3", "LOCAL_RANK": "3", "WORLD_SIZE": "8"}
    assert launch.ensure_world(8, "/x/bench.py", [], env=env, _exec=None) == (3, 3, 8, True)
    with pytest.raises(SystemExit) as e:                      # the launcher's world is not what --gpus says: loud, not a 1-GPU run
        launch.ensure_world(4, "/x/bench.py", [], env=env, _exec=None)
    assert e.value.code == 2


def test_bench_rejects_a_world_size_mismatch_before_touching_the_gpu():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29578")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, cwd=ROOT,
                         timeout=300, env=env)
    assert out.returncode == 2 and "WORLD_SIZE=2" in out.stderr


def test_rank_core_plan_is_disjoint_and_numa_local():
    allowed = list(range(32))
    node_cpus = {0: range(0, 16), 1: range(16, 32)}
    plan = launch.plan_rank_cores(allowed, 8, [0, 0, 0, 0, 1, 1, 1, 1], node_cpus)
    assert [len(p) for p in plan] == [4] * 8
    assert sorted(c for p in plan for c in p) == allowed
    assert all(c < 16 for p in plan[:4] for c in p) and all(c >= 16 for p in plan[4:] for c in p)
    plan = launch.plan_rank_cores(allowed, 4, [None] * 4, {})           # unknown topology: an equal split
    assert [len(p) for p in plan] == [8] * 4 and len({c for p in plan for c in p}) == 32
    plan = launch.plan_rank_cores(list(range(8, 24)), 2, [0, 1], node_cpus)   # affinity mask narrower than the nodes
    assert plan == [list(range(8, 16)), list(range(16, 24))]


def test_parse_cpulist():
    assert launch._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
