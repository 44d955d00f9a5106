ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp; cd $ROOT
O=gpurun_out/d2h; rm -rf $O; mkdir -p $O
cat > /tmp/d2h_probe.py <<'P'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from mcquic_amd import Compressor
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = Compressor(128, 2, [8192, 2048, 512]).eval().to(dev)
x = (torch.rand((32, 3, 768, 512)) * 2 - 1).to(dev)
yhs = [torch.empty(x.shape, pin_memory=True) for _ in range(2)]
out_stream = torch.cuda.Stream(dev)
mode = sys.argv[1]
def run(n):
    main = torch.cuda.current_stream(dev); keep = [None, None]
    for i in range(n):
        y = model.decode(model.encode(x))
        if mode == "main":
            yhs[0].copy_(y, non_blocking=True)
        else:
            ready = torch.cuda.Event(); ready.record(main)
            with torch.cuda.stream(out_stream):
                out_stream.wait_event(ready)
                yhs[i % 2].copy_(y, non_blocking=True)
            keep[i % 2] = y
    torch.cuda.synchronize()
run(2)
t0 = time.perf_counter(); run(4); print(mode, (time.perf_counter() - t0) / 4 * 1e3, "ms per batch")
P
for m in main side; do
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/$m -o t -- python /tmp/d2h_probe.py $m 2>/dev/null | grep "ms per batch"
  f=$(find $O/$m -name "*memory_copy_trace.csv" | head -1)
  echo "== $m: copies (direction, bytes, duration us)"; python - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    b = int(r.get("Bytes", r.get("bytes", 0)) or 0)
    if b > 1e8:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(r.get("Direction", r.get("Name", "?")), b, round(d, 1), "us", round(b / d / 1e3, 1), "GB/s")
P
  k=$(find $O/$m -name "*kernel_trace.csv" | head -1)
  python - "$k" <<'P'
import csv, sys, collections
tot = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Kernel_Name"][:60]; tot[nm] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; n[nm] += 1
for nm, t in tot.most_common(4): print(f"   {t/1e3:9.2f} ms {n[nm]:5d}  {nm}")
P
  rm -rf $O/$m
done
