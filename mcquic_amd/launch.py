"""One process per GPU: the launch helpers shared by bench.py and tools/bench_train.py.

The reference starts its multi-GPU jobs with `torchrun --nproc_per_node N` (singlenode.sh:21) and its throughput harness
(`Validator.speed`, mcquic/validate/validator.py:60-97) runs on whatever rank it is called on.  Here:

  * `ensure_world(gpus, script_argv)`: `--gpus N > 1` without a process group in the environment re-executes the script
    under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P`;
    a WORLD_SIZE that disagrees with `--gpus` is an error (exit code 2), never a silent 1-GPU run.
  * `pin_rank_cores(local_rank, local_world)`: every rank drives ~330 Python-side kernel launches per step from one host
    thread; ranks of one node get disjoint core sets, taken from the NUMA node of their GPU when sysfs tells.
  * `rccl_check(dist, dev)`: an actual all-reduce over the process group; its result is the world size RCCL really spans.
"""
from __future__ import annotations

import os
import socket
import sys
from typing import Dict, List, Optional, Sequence


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher_argv(gpus: int, script: str, script_args: Sequence[str], port: Optional[int] = None, python: Optional[str] = None) -> List[str]:
    """The command line that runs `script script_args...` as `gpus` ranks on this node (one per GPU)."""
    if gpus < 1:
        raise ValueError(f"--gpus must be >= 1, got {gpus}")
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port if port is not None else free_port()),
            script, *script_args]


def world_from_env(env: Optional[Dict[str, str]] = None):
    """(rank, local_rank, world, launched) from what torch.distributed.run exports."""
    env = os.environ if env is None else env
    launched = "RANK" in env and "WORLD_SIZE" in env
    return int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")), int(env.get("WORLD_SIZE", "1")), launched


def ensure_world(gpus: int, script: str, script_args: Sequence[str], env: Optional[Dict[str, str]] = None, _exec=os.execvpe):
    """Make `--gpus` and the process group agree.  Returns (rank, local_rank, world, launched) when this process is a
    rank of the right world; re-executes under torch.distributed.run when `gpus > 1` and no launcher set the
    environment (does not return then); exits with code 2 when the launcher's WORLD_SIZE is not `gpus`."""
    env = os.environ if env is None else env
    rank, local_rank, world, launched = world_from_env(env)
    if launched:
        if world != gpus:
            sys.stderr.write(f"{os.path.basename(script)}: --gpus {gpus} but the launcher started WORLD_SIZE={world} ranks\n")
            raise SystemExit(2)
        return rank, local_rank, world, True
    if gpus > 1:
        argv = launcher_argv(gpus, script, script_args)
        child_env = dict(env)
        child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
        child_env.setdefault("OMP_NUM_THREADS", "1")
        sys.stdout.flush()
        sys.stderr.flush()
        _exec(argv[0], argv, child_env)
        raise SystemExit(2)                                          # only reached with a stubbed _exec
    return rank, local_rank, world, False


def _parse_cpulist(text: str) -> List[int]:
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_node(pci_bus_id: Optional[str]) -> Optional[int]:
    """NUMA node of a GPU from sysfs (`/sys/bus/pci/devices/<id>/numa_node`), None when unknown."""
    if not pci_bus_id:
        return None
    for name in (pci_bus_id.lower(), "0000:" + pci_bus_id.lower().split(":", 1)[-1] if pci_bus_id.count(":") == 1 else None):
        if not name:
            continue
        try:
            node = int(open(f"/sys/bus/pci/devices/{name}/numa_node").read())
            return node if node >= 0 else None
        except (OSError, ValueError):
            continue
    return None


def plan_rank_cores(allowed: Sequence[int], local_world: int, numa_of_rank: Sequence[Optional[int]],
                    node_cpus: Dict[int, Sequence[int]]) -> List[List[int]]:
    """Disjoint core sets for the `local_world` ranks of a node out of the `allowed` cores.  Ranks whose GPU sits on a
    known NUMA node share that node's allowed cores among themselves; the others share what is left."""
    allowed = sorted(set(allowed))
    pools: Dict[Optional[int], List[int]] = {}
    members: Dict[Optional[int], List[int]] = {}
    for r in range(local_world):
        node = numa_of_rank[r] if r < len(numa_of_rank) else None
        cpus = [c for c in node_cpus.get(node, []) if c in set(allowed)] if node is not None else []
        key = node if cpus else None
        members.setdefault(key, []).append(r)
        if key is not None:
            pools[key] = cpus
    claimed = {c for cpus in pools.values() for c in cpus}
    pools[None] = [c for c in allowed if c not in claimed] or list(allowed)
    plan: List[List[int]] = [[] for _ in range(local_world)]
    for key, ranks in members.items():
        cpus = pools[key]
        per = len(cpus) // len(ranks)
        for i, r in enumerate(ranks):
            plan[r] = cpus[i * per:(i + 1) * per] if per >= 1 else list(cpus)
    return plan


def pin_rank_cores(local_rank: int, local_world: int) -> Optional[List[int]]:
    """Pin this rank to its share of the node's cores (no-op at local_world == 1 or without sched_setaffinity)."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity") or os.environ.get("MCQUIC_AMD_PIN_CORES", "1") == "0":
        return None
    try:
        import torch
        numa = []
        for i in range(local_world):
            try:
                prop = torch.cuda.get_device_properties(i)
                bus = getattr(prop, "pci_bus_id", None)
                dom = getattr(prop, "pci_domain_id", 0)
                devid = getattr(prop, "pci_device_id", 0)
                numa.append(gpu_numa_node(f"{dom:04x}:{bus:02x}:{devid:02x}.0") if bus is not None else None)
            except Exception:            # noqa: BLE001 -- best effort: unknown topology = equal split
                numa.append(None)
        node_cpus = {}
        for node in {n for n in numa if n is not None}:
            try:
                node_cpus[node] = _parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())
            except OSError:
                pass
        plan = plan_rank_cores(sorted(os.sched_getaffinity(0)), local_world, numa, node_cpus)
        mine = plan[local_rank]
        if len(mine) >= 2:               # a rank needs its launch thread plus RCCL's proxy thread
            os.sched_setaffinity(0, mine)
            return mine
    except Exception:                    # noqa: BLE001 -- pinning is an optimisation: whatever sysfs / the runtime says, run unpinned
        pass
    return None


def rccl_check(dist, dev) -> int:
    """All-reduce one float per rank over the process group; returns the sum = the number of ranks RCCL connected."""
    import torch
    one = torch.ones(1, dtype=torch.float32, device=dev)
    dist.all_reduce(one)
    return int(round(float(one.item())))
