"""Where does autograd itself add gradients in the training step?  A tensor consumed by two nodes gets its gradient from an
ATen `add` the engine issues (one launch per extra consumer, not ours).  Walks the graph of one training forward and prints every
(producer node, output index) that more than one consumer points at, with the consumers' names and the tensor's size.
    python tools/autograd_fanin.py          (on the GPU box)
"""
import collections
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["MCQUIC_AMD_BRANCH_STREAMS"] = "0"
import torch

from mcquic_amd import Compressor
from mcquic_amd.autograd import mse_loss


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(3407)
    if "--neon" in sys.argv or "--neon-dense" in sys.argv:
        from mcquic_amd import Neon
        model = Neon(32, 256, [8, 4, 2, 2], "--neon-dense" in sys.argv).to(dev).train()
        x = (torch.rand((2, 3, 128, 128)) * 2 - 1).to(dev)
    else:
        model = Compressor(128, 2, [8192, 2048, 512]).to(dev).train()
        x = (torch.rand((8, 3, 256, 256)) * 2 - 1).to(dev)
    xHat = model(x)[0]
    loss = mse_loss(xHat, x)
    consumers = collections.defaultdict(list)
    seen, stack = set(), [loss.grad_fn]
    while stack:
        fn = stack.pop()
        if fn is None or id(fn) in seen:
            continue
        seen.add(id(fn))
        for nxt, idx in fn.next_functions:
            if nxt is None:
                continue
            consumers[(id(nxt), idx)].append(type(fn).__name__)
            consumers[(id(nxt), idx, "node")] = nxt
            stack.append(nxt)
    rows = []
    for key, names in consumers.items():
        if len(key) == 2 and len(names) > 1:
            node = consumers[(key[0], key[1], "node")]
            if type(node).__name__ == "AccumulateGrad":
                v = node.variable
                rows.append((f"AccumulateGrad{tuple(v.shape)}", key[1], names))
            else:
                rows.append((type(node).__name__, key[1], names))
    print(f"# {len(seen)} nodes; {len(rows)} outputs with more than one consumer (each extra consumer = one engine-issued add)")
    for name, idx, names in sorted(rows, key=lambda r: r[0]):
        print(f"{name}[{idx}] <- {', '.join(sorted(names))}")


if __name__ == "__main__":
    main()
