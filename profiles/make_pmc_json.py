#!/usr/bin/env python3
"""profiles/pmc_stats.py writes /tmp/pmc_rows.json (per kernel x launch grid averages of the --pmc passes); this turns
it into the summary bench.py reads for `roofline.traffic` (profiles/rNN_pmc.json).
usage: python profiles/make_pmc_json.py gpurun_out/r02/pmc_rows.json > profiles/rNN_pmc.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcquic_amd.build import csrc_sha          # noqa: E402

rows = json.load(open(sys.argv[1]))
dom = max((r for r in rows if "conv_mfma" in r["kernel"]), key=lambda r: r["_dur_ns"] * r["launches"])["kernel"]
sel = [r for r in rows if r["kernel"] == dom]
n = sum(r["launches"] for r in sel)
fetch = sum(r["FETCH_SIZE"] * 1024 * r["launches"] for r in sel) / n
write = sum(r["WRITE_SIZE"] * 1024 * r["launches"] for r in sel) / n
big = next(r for r in sel if r["wgs"] == 3072 and r["gy"] == 1)          # the 192x128 level, 128 -> 128 channels
large = [r for r in sel if r["wgs"] >= 3072]
busy = sum(r["SQ_VALU_MFMA_BUSY_CYCLES"] * r["launches"] for r in large)
gui = sum(r["GRBM_GUI_ACTIVE"] * r["launches"] for r in large)
dur = sum(r["_dur_ns"] * r["launches"] for r in large)
allc = [r for r in rows if "conv_mfma" in r["kernel"] or "conv_head16" in r["kernel"]]
na = sum(r["launches"] for r in allc)
out = {
    "csrc_sha": csrc_sha(),      # the kernel sources these passes ran on; bench.py prints traffic_stale when its own differ
    "source": "rocprofv3 --pmc {FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE} --kernel-trace -- "
              "python bench.py --steps 1 --warmup 1 --no-cpu-baseline (three separate passes, single stream; "
              "tools/collect.sh, profiles/pmc_stats.py, profiles/make_pmc_json.py)",
    "kernel": dom + " (128 co x 64 px wave tile, 3x3, no prologue)",
    "units": "bytes per launch, averaged over the launches of the kernel in the run; FETCH_SIZE/WRITE_SIZE are KiB "
             "counters; on gfx950 FETCH_SIZE tallies 64 B per 128-B request for wide coalesced reads "
             "(MI355X_MICROARCH.md, HBM section), hence the x2-corrected figure (calibrated, see `calibration`)",
    "launches_summed_over_the_passes": n,
    "fetch_size_bytes_per_launch": fetch,
    "write_size_bytes_per_launch": write,
    "fetch_bytes_per_launch_x2_corrected": 2 * fetch,
    "fetch_size_bytes_192x128_layer": big["FETCH_SIZE"] * 1024,
    "fetch_bytes_192x128_layer_x2_corrected": 2 * big["FETCH_SIZE"] * 1024,
    "write_size_bytes_192x128_layer": big["WRITE_SIZE"] * 1024,
    "avg_us_192x128_layer": big["_dur_ns"] / 1e3,
    # SQ_VALU_MFMA_BUSY_CYCLES sums over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
    "mfma_pipe_utilisation_large_layers": busy / (gui / 8 * 1024),
    "effective_clock_ghz_large_layers": (gui / 8) / dur,
    "all_conv_launches": {
        "note": "every conv_mfma_kernel / conv_head16_kernel instance, averaged per launch (the population bench.py averages its algorithmic bytes over)",
        "fetch_size_bytes_per_launch": sum(r["FETCH_SIZE"] * 1024 * r["launches"] for r in allc) / na,
        "fetch_bytes_per_launch_x2_corrected": 2 * sum(r["FETCH_SIZE"] * 1024 * r["launches"] for r in allc) / na,
        "write_size_bytes_per_launch": sum(r["WRITE_SIZE"] * 1024 * r["launches"] for r in allc) / na,
    },
    "calibration": "tools/probe_fetch_size.py: FETCH_SIZE reads 0.500 of the known bytes for float4 global loads AND for the conv "
                   "kernel's per-lane buffer_load_dword (1x1 conv streaming 1.6 GB once), WRITE_SIZE reads 1.000: the x2 correction applies",
    "algorithmic_bytes_192x128_layer": {"input": 402653184, "output": 402653184,
                                         "note": "+ one more output-sized read (residual) and write (SiLU twin) on the closing conv of a block"},
}
json.dump(out, sys.stdout, indent=1)
