#!/usr/bin/env python3
"""In-kernel duration of small conv launches (kernel tuning aid).

    rocprofv3 --kernel-trace -d gpurun_out/ps -o ps -- python tools/probe_small.py run
    python tools/probe_small.py report gpurun_out/ps/*/ps_results.db      (or wherever rocprofv3 put the rocpd file)

`run` launches every configuration REPS times back to back; `report` reads the dispatch durations in order and prints
the median per configuration (rocprof durations exclude the launch gaps that event timing includes).
"""
import glob
import os
import sqlite3
import sys

REPS = 24
CONFIGS = []          # (n, cin, h, w, tile, flags)
if os.environ.get("PROBE_SHAPES"):      # e.g. PROBE_SHAPES="8,16,16;8,8,8" PROBE_TILES="0,0x11,0x311,0x241"
    tiles = [int(t, 0) for t in os.environ.get("PROBE_TILES", "0,0x311,0x211,0x241,0x341,0x222,0x322,0x221,0x321,0x121").split(",")]
    for shp in os.environ["PROBE_SHAPES"].split(";"):
        n, h, w = (int(v) for v in shp.split(","))
        for tile in tiles:
            CONFIGS.append((n, 128, h, w, tile, "res"))
else:
    for n in (1, 32):
        for (h, w) in ((12, 8), (24, 16), (48, 32)):
            for cin in (16, 128):
                for tile in (0, 0x42, 0x11, 0x311, 0x241):
                    for flags in ("plain", "res"):
                        CONFIGS.append((n, cin, h, w, tile, flags))


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mcquic_amd import ops
    dev = torch.device("cuda:0")
    cout = 128
    for (n, cin, h, w, tile, flags) in CONFIGS:
        x = torch.randn(n, cin, h, w, device=dev)
        res = torch.randn(n, cout, h, w, device=dev)
        pack = ops.PackedConv(torch.randn(cout, cin, 3, 3, device=dev) * 0.03, torch.randn(cout, device=dev))
        kw = dict(tile=tile)
        if flags == "res":
            kw.update(res=res, dual_silu=True)
        torch.cuda.synchronize()
        for _ in range(REPS):
            ops.conv2d(x, pack, 1, **kw)
        torch.cuda.synchronize()


def report(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, end - start from kernels where name like '%conv_mfma_kernel%' order by start").fetchall()
    assert len(rows) == REPS * len(CONFIGS), (len(rows), REPS * len(CONFIGS))
    print(f"{'n':>3} {'cin':>4} {'hxw':>6} {'tile':>6} {'flags':>6} {'median_us':>10} {'min_us':>8}  kernel")
    for i, cfg in enumerate(CONFIGS):
        d = sorted(r[1] / 1e3 for r in rows[i * REPS:(i + 1) * REPS])
        nm = rows[i * REPS][0].replace("(anonymous namespace)::", "").replace("void ", "")[:40]
        n, cin, h, w, tile, flags = cfg
        print(f"{n:3d} {cin:4d} {h:3d}x{w:<3d} {tile:#6x} {flags:>6} {d[len(d) // 2]:10.1f} {d[0]:8.1f}  {nm}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2] if len(sys.argv) > 2 else sorted(glob.glob("gpurun_out/ps/**/*_results.db", recursive=True))[-1])
