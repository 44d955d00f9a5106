"""Build recipe for libmcquic_hip.so (gfx950 only, hipcc; no cmake, no JIT cache -- the .so lives in-tree)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmcquic_hip.so")
SOURCES = ["conv_mfma.hip", "vq.hip", "vq_train.hip", "train_ops.hip", "metrics.hip", "rans.cpp"]
HEADERS = ["mcq_common.h", "vq_common.h", "conv_head16.h", os.path.join("..", "..", "include", "mcquic_hip.h")]
# -ffp-contract=off: element-wise epilogues keep the reference's one-rounding-per-op sequence
#   (e.g. a * sigmoid(b) then + x are two torch kernels in mcquic/nn/blocks.py:286-287).
# -pragma-unroll-threshold: the 128-register epilogue must be fully unrolled or the accumulators spill to scratch.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-mllvm", "-pragma-unroll-threshold=1000000"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip into mcquic_amd/libmcquic_hip.so with hipcc (cross-compiles without a GPU)."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libmcquic_hip.so cannot be built")
    cmd = [hipcc] + FLAGS + ["-o", LIB + ".tmp"] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
