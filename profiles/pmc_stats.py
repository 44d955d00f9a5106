#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (rocpd sqlite) per kernel / launch grid.
usage: python profiles/pmc_stats.py gpurun_out/pmc_FETCH_SIZE/pmc_results.db gpurun_out/pmc_WRITE_SIZE/pmc_results.db ...
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide
coalesced reads (MI355X_MICROARCH.md, HBM section), so the corrected read bytes are 2 x FETCH_SIZE x 1024."""
import json
import os
import sqlite3
import sys
from collections import defaultdict

# kernels summarised: name substrings, PMC_KERNELS=a,b,c overrides (the training step adds the weight-gradient and VQ-training kernels)
WANT = [w for w in os.environ.get("PMC_KERNELS", "conv_mfma,conv_head16,vq_assign").split(",") if w]

agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in sys.argv[1:]:
    cur = sqlite3.connect(path).cursor()
    # one row per (dispatch, counter instance): sum the instances of a dispatch first
    per = defaultdict(float)
    meta = {}
    for did, name, gx, gy, wx, cname, val, dur in cur.execute(
            "select dispatch_id, kernel_name, grid_size_x, grid_size_y, workgroup_size_x, counter_name, value, duration from counters_collection"):
        if not any(w in name for w in WANT):
            continue
        per[(did, cname)] += val
        meta[did] = (name, gx, gy, wx, dur)
    seen = set()
    for (did, cname), val in per.items():
        name, gx, gy, wx, dur = meta[did]
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        key = (short, gx // max(wx, 1), gy)
        a = agg[key][cname]
        a[0] += val
        a[1] += 1
        if did not in seen:
            seen.add(did)
            d = agg[key]["_dur_ns"]
            d[0] += dur
            d[1] += 1
rows = []
for key, cs in agg.items():
    r = {"kernel": key[0], "wgs": key[1], "gy": key[2]}
    for c, (s, n) in cs.items():
        r[c] = s / n
        r["launches"] = max(r.get("launches", 0), n)
    rows.append(r)
rows.sort(key=lambda r: -r.get("_dur_ns", 0) * r.get("launches", 0))
print(f"{'kernel':34s} {'wgs':>6} {'gy':>2} {'n':>4} {'avg_us':>9} {'read_MB(x2 corr)':>17} {'write_MB':>9} {'mfma_busy/(gui*1024)':>21}")
for r in rows[:24]:
    rd = 2 * r.get("FETCH_SIZE", float("nan")) * 1024 / 1e6
    wr = r.get("WRITE_SIZE", float("nan")) * 1024 / 1e6
    util = r.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")) / (r.get("GRBM_GUI_ACTIVE", float("nan")) * 1024)
    print(f"{r['kernel']:34s} {r['wgs']:6d} {r['gy']:2d} {r['launches']:4d} {r.get('_dur_ns', 0) / 1e3:9.1f} {rd:17.1f} {wr:9.1f} {util:21.3f}")
# any further counter of the passes (e.g. SQ_INSTS_VALU, SQ_INSTS_MFMA: instruction counts per launch, summed over the chip)
KNOWN = {"FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "_dur_ns", "kernel", "wgs", "gy", "launches"}
extra = sorted({c for r in rows for c in r if c not in KNOWN})
if extra:
    print()
    print(f"{'kernel':34s} {'wgs':>6} {'gy':>2} " + " ".join(f"{c:>28s}" for c in extra))
    for r in rows[:24]:
        print(f"{r['kernel']:34s} {r['wgs']:6d} {r['gy']:2d} " + " ".join(f"{r.get(c, float('nan')):28.0f}" for c in extra))
json.dump(rows, open("/tmp/pmc_rows.json", "w"))
