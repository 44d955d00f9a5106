"""Child process of tests/test_gpu_step_ops.py::test_training_step_replay_issues_no_aten_kernels: captures one training step
(forward + backward) of a small Compressor as a hipGraph, replays it under torch.profiler and prints the names of the device
activities of that replay as one JSON list."""
import json
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ["MCQUIC_AMD_BRANCH_STREAMS"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from mcquic_amd import Compressor  # noqa: E402
from mcquic_amd.autograd import backward, mse_loss  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(3407)
    model = Compressor(16, 2, [64, 32, 16]).to(dev).train()
    x = (torch.rand((2, 3, 128, 128), generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)

    def step():
        for p in model.parameters():
            p.grad = None
        xHat, _, _, _ = model(x)
        loss = mse_loss(xHat, x)
        backward(loss)
        return loss
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    for p in model.parameters():
        p.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    graph.replay()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        graph.replay()
        torch.cuda.synchronize()
    names = []
    for e in prof.profiler.kineto_results.events():
        dt = str(e.device_type())
        if "CUDA" in dt or "HIP" in dt or "PrivateUse" in dt:
            names.append(e.name())
    print(json.dumps(names))


if __name__ == "__main__":
    main()
