#!/usr/bin/env python3
"""Second step of the batch-1 discrepancy hunt (tools/probes/batch1_order.py found: a graph captured AFTER eager 32-image steps
replays at 6.67 ms, a graph captured in a fresh process at 5.90, eager at 5.86 either way).  Run under rocprofv3 --kernel-trace:
phases separated by 0.4 s of idle; `--analyse DB` then prints per phase the span, the sum of kernel durations and the kernels that
changed most between the phases.

    phase 0  fresh capture, 20 replays
    phase 1  (after 3 eager B32 steps) the SAME graphs again, 20 replays
    phase 2  graphs captured now, 20 replays
    phase 3  eager batch 1, 20 steps"""
import os
import sqlite3
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def analyse(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    phases, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[1] - cur[-1][2] > 200e6:            # > 0.2 s of idle
            phases.append(cur)
            cur = []
        cur.append(r)
    phases.append(cur)
    print(f"{len(phases)} phases")
    per = []
    for i, ph in enumerate(phases):
        span = (ph[-1][2] - ph[0][1]) / 1e6
        busy = sum(e - s for _, s, e in ph) / 1e6
        names = {}
        for n, s, e in ph:
            k = n.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
            c = names.setdefault(k, [0, 0.0])
            c[0] += 1
            c[1] += (e - s) / 1e3
        per.append(names)
        print(f"phase {i}: {len(ph)} kernels, span {span:.2f} ms, kernel time {busy:.2f} ms")
    for a, b in ((1, 3), (1, 5), (1, 6)):
        if max(a, b) >= len(per):
            continue
        print(f"--- phase {a} vs phase {b}: kernels by change of total time (us per replay, /20)")
        keys = set(per[a]) | set(per[b])
        diff = sorted(((per[b].get(k, [0, 0])[1] - per[a].get(k, [0, 0])[1]) / 20, k) for k in keys)
        for d, k in diff[:6] + diff[-12:]:
            print(f"  {d:9.1f}  {per[a].get(k, [0, 0])[0] / 20:6.1f} -> {per[b].get(k, [0, 0])[0] / 20:6.1f} calls  "
                  f"{per[a].get(k, [0, 0])[1] / 20:9.1f} -> {per[b].get(k, [0, 0])[1] / 20:9.1f} us  {k}")


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
        return analyse(sys.argv[2])
    import torch
    from mcquic_amd.utils import synthetic
    dev = torch.device("cuda", 0)
    model = synthetic.bench_model().to(dev)
    x = synthetic.bench_images(0, 32).to(dev)
    x1 = x[:1].contiguous()

    def b1(n=20):
        for _ in range(n):
            model.decode(model.encode(x1))
        torch.cuda.synchronize()
        time.sleep(0.4)
    model.enableGraphs(True)
    b1(3)                        # phase 0: capture + warm replays
    b1()                         # phase 1: fresh graphs
    graphs, stamp = model._graphs, model._graphStamp
    model.enableGraphs(False)
    for _ in range(3):
        model.decode(model.encode(x))
    torch.cuda.synchronize()
    time.sleep(0.4)              # phase 2: the B32 steps
    model._graphs, model._graphStamp = graphs, stamp
    b1()                         # phase 3: the OLD graphs after B32
    model.enableGraphs(True)
    b1(3)                        # phase 4: new capture
    b1()                         # phase 5: new graphs
    model.enableGraphs(False)
    b1()                         # phase 6: eager


if __name__ == "__main__":
    main()
