#!/usr/bin/env python3
"""Ordered kernel sequence of the LAST captured-step replay in a rocprofv3 --kernel-trace database (tools/bench_train.py --graph):
    python tools/probes/dump_step_sequence.py <kt_results.db> > sequence.txt
Replays are separated by looking for the step's first kernel name recurring; prints index, start offset (us), duration (us), name."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, grid_x, grid_y, grid_z from kernels order by start").fetchall()
names = [r[0] for r in rows]
# the last replay: skip the run's epilogue (the bench's gradient-norm checksum), then find the period of the replays before it
first = end = None
for skip in range(0, 8000, 7):
    body = rows[:len(rows) - skip] if skip else rows
    for period in range(300, 3000):
        if len(body) > 2 * period and "conv_mfma" in body[-1][0] + body[-2][0] + body[-3][0] + body[-40][0] and all(body[-1 - i][0] == body[-1 - i - period][0] for i in range(period)):
            first, end = len(body) - period, len(body)
            break
    if first is not None:
        break
if first is None:
    raise SystemExit("no periodic stretch found")
tail = rows[:end]
seq = tail[first:]
t0 = seq[0][1]
print(f"# {len(seq)} kernels per replay, span {(seq[-1][2] - t0) / 1e3:.1f} us, kernel time {sum(r[2] - r[1] for r in seq) / 1e3:.1f} us")
for i, (n, s, e, gx, gy, gz) in enumerate(seq):
    short = n.replace("(anonymous namespace)::", "").replace("void ", "")[:90]
    print(f"{i:5d} {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  {gx}x{gy}x{gz}  {short}")
