"""Smallest graph with the pattern of ops.rng_snapshot: D2D memcpy node (clone of an int64[2] state) followed by a kernel that
advances the state, three times per replay.  Prints state and snapshots after every replay; `nsync` host synchronisations between
replays.  Expected: offsets 3i, snapshots (3i, 3i+1, 3i+2)."""
import sys, torch
dev = torch.device("cuda:0")
nsync = int(sys.argv[1]) if len(sys.argv) > 1 else 666
kernel_copy = len(sys.argv) > 2 and sys.argv[2] == "k"
pad = int(sys.argv[3]) if len(sys.argv) > 3 else 0
st = torch.tensor([42, 0], dtype=torch.int64).to(dev)
junk = torch.randn(1 << 20, device=dev)
trig = sys.argv[4] if len(sys.argv) > 4 else "n"
ts = [torch.ones(1000 + 37 * i, device=dev) for i in range(666)]
big_src = torch.arange(1 << 22, device=dev, dtype=torch.float32)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())


def body():
    snaps = []
    acc = junk
    for _ in range(3):
        for _ in range(pad):                      # kernels around the copies, like the training step has
            acc = acc * 1.0001
        snap = torch.add(st, 0) if kernel_copy else st.clone()
        st[1:].add_(1)
        snaps.append(snap)
    return torch.stack(snaps), big_src.clone()


with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s)
st[1] = 0
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out, acc = body()
bad = 0
for i in range(12):
    g.replay()
    for _ in range(nsync):
        torch.cuda.synchronize()
    if trig == "c":                                # eager reductions + D2H copies (what broke memset nodes, memset_node_probe.py)
        sum(int(not torch.isfinite(t).all()) for t in ts)
    torch.cuda.synchronize()
    got = out[:, 1].tolist()
    ok = got == [3 * i, 3 * i + 1, 3 * i + 2] and int(st[1]) == 3 * i + 3 and torch.equal(acc, big_src)
    big_src.add_(1.0)
    bad += not ok
    print(i, "state", int(st[1]), "snapshots", got, "" if ok else "  <-- WRONG", flush=True)
print("kernel copy" if kernel_copy else "memcpy node", "nsync", nsync, "pad", pad, "wrong replays:", bad)
