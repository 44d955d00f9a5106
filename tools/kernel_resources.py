#!/usr/bin/env python3
"""Registers / scratch / occupancy of every kernel in a HIP source, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    python tools/kernel_resources.py mcquic_amd/csrc/conv_mfma.hip [extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcquic_amd import build as B  # noqa: E402


def main():
    src = os.path.abspath(sys.argv[1])
    cmd = ["hipcc"] + B.CFLAGS + sys.argv[2:] + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    run = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(src))
    txt = run.stderr
    if run.returncode != 0:                      # (a failed compile must not look like "no kernels")
        sys.stderr.write("".join(ln + "\n" for ln in txt.splitlines() if "error" in ln or "note:" in ln)[:4000])
        sys.exit(run.returncode)
    keys = {"vgpr": r"    VGPRs", "agpr": r"AGPRs", "scratch": r"ScratchSize \[bytes/lane\]", "occ": r"Occupancy \[waves/SIMD\]",
            "spill": r"VGPRs Spill", "lds": r"LDS Size \[bytes/block\]"}
    for b in re.split(r"remark: Function Name: ", txt)[1:]:
        name = b.split(" ")[0]
        d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
        vals = {}
        for k, pat in keys.items():
            m = re.search(pat + r": (\d+)", b)
            vals[k] = int(m.group(1)) if m else -1
        print(f"{d[:66]:66s} vgpr {vals['vgpr']:4d} agpr {vals['agpr']:4d} scratch {vals['scratch']:5d} occ {vals['occ']} spill {vals['spill']:4d} lds {vals['lds']}")


if __name__ == "__main__":
    main()
