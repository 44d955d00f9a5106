#!/usr/bin/env python3
"""A few eager steps of Neon at the trainer's shapes for a kernel trace (tools/kt.sh):
    tools/kt.sh neon_train python tools/prof_neon.py --train [--dense]      # forward + backward, 4 x 512x512
    tools/kt.sh neon_infer python tools/prof_neon.py [--dense]              # encode + decode, 8 x 512x512"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MCQUIC_AMD_BRANCH_STREAMS"] = "0"
import torch  # noqa: E402

from mcquic_amd import Neon  # noqa: E402
from mcquic_amd.autograd import backward, mse_loss  # noqa: E402


def main():
    dense, train = "--dense" in sys.argv, "--train" in sys.argv
    dev = torch.device("cuda:0")
    torch.manual_seed(3407)
    model = Neon(32, 4096, [16, 8, 4, 2, 2], dense).to(dev)
    x = (torch.rand((4 if train else 8, 3, 512, 512), generator=torch.Generator().manual_seed(1)) * 2 - 1).to(dev)
    if train:
        model.train()
        for _ in range(4):
            for p in model.parameters():
                p.grad = None
            backward(mse_loss(model(x)[0], x))
    else:
        model.eval()
        for _ in range(4):
            model.decode(model.encode(x))
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
