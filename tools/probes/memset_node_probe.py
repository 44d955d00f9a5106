"""torch-only: a captured ATen reduction large enough for the two-stage (global) path -- whose semaphores are zeroed by a
hipMemsetAsync, i.e. a memset NODE -- replayed with eager work between the replays.  The input grows by 1 before every replay, so
a stale output is visible.  argv[1]: trigger (n = none, s = 666 synchronize, c = 666 x int(isfinite(t).all()) on small tensors);
argv[2]: number of kernels of padding in the graph before the reduction."""
import sys, torch
dev = torch.device("cuda:0")
trig = sys.argv[1] if len(sys.argv) > 1 else "c"
pad = int(sys.argv[2]) if len(sys.argv) > 2 else 0
big = torch.rand(8 * 3 * 256 * 256, device=dev)
other = torch.rand_like(big)
ts = [torch.ones(1000 + 37 * i, device=dev) for i in range(666)]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())


def body():
    a = big
    for _ in range(pad):
        a = a * 1.0
    return torch.nn.functional.mse_loss(a, other), a.sum()


with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out, tot = body()
bad = 0
for i in range(10):
    big.add_(1.0)
    g.replay()
    if trig == "s":
        for _ in range(666):
            torch.cuda.synchronize()
    if trig == "c":
        sum(int(not torch.isfinite(t).all()) for t in ts)
    torch.cuda.synchronize()
    want, want2 = float(torch.nn.functional.mse_loss(big, other)), float(big.sum())
    ok = abs(float(out) - want) <= 1e-4 * want and abs(float(tot) - want2) <= 1e-4 * want2
    bad += not ok
    print(i, "graph", float(out), float(tot), "eager", want, want2, "" if ok else "  <-- STALE", flush=True)
print("trigger", trig, "pad", pad, "stale replays:", bad)
