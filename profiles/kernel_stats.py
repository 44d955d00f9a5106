#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite `*_results.db`) per kernel and per launch grid.
usage: python profiles/kernel_stats.py gpurun_out/prof/bench_results.db > profiles/rNN_kernel_stats.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                   "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# per kernel: total {tot:.2f} ms over {sum(r[1] for r in rows)} dispatches")
print(f"{'total_ms':>10} {'pct':>6} {'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10}  kernel")
for r in rows:
    print(f"{r[2]:10.2f} {100 * r[2] / tot:6.2f} {r[1]:6d} {r[3]:10.1f} {r[4]:10.1f} {r[5]:10.1f}  {r[0][:120]}")
print("\n# conv / VQ / weight-gradient kernels by launch grid (workgroups x grid.y)")
rows = cur.execute("select name, grid_x, grid_y, workgroup_x, count(*), avg(end-start)/1e3, min(end-start)/1e3, sum(end-start)/1e6, "
                   "max(vgpr_count), max(accum_vgpr_count), grid_z from kernels where name like '%conv_mfma%' or name like '%conv_head16%' or name like '%conv_t16%' or name like '%conv_r16%' or name like '%vq_%' or name like '%wgrad%' "
                   "or name like '%fused_%' group by name, grid_x, grid_y, grid_z order by 8 desc").fetchall()
print(f"{'total_ms':>10} {'calls':>6} {'avg_us':>10} {'min_us':>10} {'wgs':>8} {'gy':>3} {'gz':>3} {'vgpr':>5}  kernel")
for r in rows:
    nm = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{r[7]:10.2f} {r[4]:6d} {r[5]:10.1f} {r[6]:10.1f} {r[1] // max(r[3], 1):8d} {r[2]:3d} {r[10]:3d} {r[8]:5d}  {nm[:60]}")
