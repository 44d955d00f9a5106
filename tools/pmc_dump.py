import sqlite3, sys
from collections import defaultdict
for path in sys.argv[1:]:
    cur = sqlite3.connect(path).cursor()
    per = defaultdict(float); meta = {}
    for did, name, gx, wx, cname, val, dur in cur.execute("select dispatch_id, kernel_name, grid_size_x, workgroup_size_x, counter_name, value, duration from counters_collection"):
        if "conv_" not in name: continue
        per[(did, cname)] += val; meta[did] = (name.replace("(anonymous namespace)::","").replace("void ","").split("(")[0], gx // max(wx,1), dur)
    agg = defaultdict(lambda: defaultdict(list))
    for (did, cname), v in per.items():
        agg[meta[did][:2]][cname].append(v)
    for did, m in meta.items():
        agg[m[:2]]["dur_us"].append(m[2] / 1e3)
    for key, cs in agg.items():
        n = len(cs["dur_us"])
        if n < 5: continue
        print(key, "n", n, {c: round(sum(v) / len(v), 1) for c, v in cs.items()})
