"""Build recipe for libmcquic_hip.so (gfx950 only, hipcc; no cmake, no JIT cache -- the .so lives in-tree).

Every source is compiled to its own object (in parallel, only when it or a header changed) and the objects are linked
into mcquic_amd/libmcquic_hip.so.  Objects live in mcquic_amd/_obj/ (git-ignored)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libmcquic_hip.so")
SOURCES = ["conv_mfma.hip", "conv_wino16.hip", "vq.hip", "vq_train.hip", "vq_bwd_mfma.hip", "train_ops.hip", "step_ops.hip", "wgrad_rows.hip", "metrics.hip", "norm.hip", "rans.cpp"]
HEADERS = ["mcq_common.h", "vq_common.h", "conv_head16.h", "conv_t16.h", "conv_wino16.h", "vq_bwd_mfma.h", "wgrad_t16.h", os.path.join("..", "..", "include", "mcquic_hip.h")]
# -ffp-contract=off: element-wise epilogues keep the reference's one-rounding-per-op sequence
#   (e.g. a * sigmoid(b) then + x are two torch kernels in mcquic/nn/blocks.py:286-287).
# -pragma-unroll-threshold: the 128-register epilogue must be fully unrolled or the accumulators spill to scratch.
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-pthread",
          "-mllvm", "-pragma-unroll-threshold=1000000"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"]


def csrc_sha() -> str:
    """SHA-256 over the sources of the dominant kernel (csrc/conv_mfma.hip and the headers it includes, names + contents): what
    measurement artefacts about that kernel are stamped with (profiles/rNN_pmc.json) so that a number collected on an older
    kernel can be recognised as stale."""
    import hashlib
    h = hashlib.sha256()
    for f in ("conv_mfma.hip", "conv_head16.h", "conv_t16.h", "mcq_common.h", "conv_wino16.hip", "conv_wino16.h"):
        h.update(f.encode() + b"\0")
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()


def _headers():
    hs = [os.path.join(CSRC, h) for h in HEADERS]
    hs += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h") and os.path.join(CSRC, f) not in hs]
    return hs + [os.path.abspath(__file__)]


def _obj_of(src: str) -> str:
    return os.path.join(OBJ, os.path.splitext(src)[0] + ".o")


def _obj_stale(src: str) -> bool:
    o = _obj_of(src)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(d) > t for d in [os.path.join(CSRC, src)] + _headers())


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + _headers()
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=(), lib: str = LIB) -> str:
    """Compile csrc/* into mcquic_amd/libmcquic_hip.so with hipcc (cross-compiles without a GPU).  `extra_flags` /
    `lib`: kernel A/B variants (e.g. -DMCQ_PFB=36 into mcquic_amd/variants/...), always compiled from scratch."""
    variant = bool(extra_flags) or lib != LIB
    if not force and not variant and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libmcquic_hip.so cannot be built")
    objdir = OBJ if not variant else lib + ".obj"
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src: str) -> str:
        o = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        if force or variant or _obj_stale(src):
            cmd = [hipcc] + CFLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True, cwd=CSRC)
        return o

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc] + LDFLAGS + ["-o", lib + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(lib + ".tmp", lib)
    if variant:
        shutil.rmtree(objdir, ignore_errors=True)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
