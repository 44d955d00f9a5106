#!/usr/bin/env python3
"""Scratch build of conv_mfma.hip with s_memrealtime stamps for tools/probes/tiny_stamps.py (round-3 probe; nothing of it ships).

Patches STAMP() lines into a COPY of the kernel source under mcquic_amd/variants/stamps/ (git-ignored), compiles that one object and
links it with the library's other objects into mcquic_amd/variants/stamps.so.  Waves 0 and 7 of every workgroup write six 100 MHz
time stamps (entry, geometry done, rings issued, k-loop done, split-K reduction done, epilogue done) into the buffer passed in the
otherwise unused gate_id pointer.

    python -m mcquic_amd.build && python tools/probes/make_stamped_conv.py
    MCQUIC_AMD_LIB=$PWD/mcquic_amd/variants/stamps.so python tools/probes/tiny_stamps.py        (on the GPU)
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mcquic_amd import build as B  # noqa: E402

s = open(os.path.join(B.CSRC, "conv_mfma.hip")).read()


def before(anchor, text):
    global s
    assert s.count(anchor) >= 1, anchor[:70]
    s = s.replace(anchor, text + anchor, 1)


def after(anchor, text):
    global s
    assert s.count(anchor) >= 1, anchor[:70]
    s = s.replace(anchor, anchor + text, 1)


before("template <int MB, int NB, int PRO, int PFA, int PFB, int TAPS, int OCC>\n__global__",
       "#define STAMP(i) do { if (stamp_buf && lane == 0 && (wave == 0 || wave == 7)) stamp_buf[(size_t)((blockIdx.z * gridDim.y + "
       "blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (wave ? 8 : 0) + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)\n")
after("    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));\n",
      "    unsigned long long* stamp_buf = (!(p.flags & MCQ_CONV_GATE) && p.gid) ? (unsigned long long*)p.gid : nullptr;\n    STAMP(0);\n")
before("    // ---- operand prefetch rings ----", "    STAMP(1);\n")
before("    // (WINO) the transformed inputs of a group of PG k-steps from its PG loads", "    STAMP(2);\n")
before("    if (WASM) asm volatile(\"s_nop 15\\n\\ts_nop 15\");", "    STAMP(3);\n")
before("        run_epilogue(active && kslice == 0, [&](int, int nb, float (&v)[16]) {", "        STAMP(4);\n")      # one-band tiles: after the barrier
before("        return;\n    }\n    float own[NB][16];", "        STAMP(5);\n")
before("    run_epilogue(active && kslice < MB, [&](int, int nb, float (&v)[16]) {", "    STAMP(4);\n")
after("        for (int r = 0; r < 16; ++r) v[r] = own[nb][r];\n    }, kslice, std::integral_constant<int, 1>{});\n", "    STAMP(5);\n")

out = os.path.join(ROOT, "mcquic_amd", "variants", "stamps")
os.makedirs(out, exist_ok=True)
open(os.path.join(out, "conv_mfma.hip"), "w").write(s)
obj = os.path.join(out, "conv_mfma.o")
subprocess.run(["hipcc"] + B.CFLAGS + ["-I" + B.CSRC, "-c", os.path.join(out, "conv_mfma.hip"), "-o", obj], check=True, cwd=out)
others = [o for o in glob.glob(os.path.join(B.OBJ, "*.o")) if os.path.basename(o) != "conv_mfma.o"]
lib = os.path.join(ROOT, "mcquic_amd", "variants", "stamps.so")
subprocess.run(["hipcc"] + B.LDFLAGS + [obj] + others + ["-o", lib], check=True)
print(lib)
