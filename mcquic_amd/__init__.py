"""mcquic_amd: MI355X-native encode/decode hot path of McQuic's Compressor (hand-written gfx950 HIP kernels
behind the reference's `mcquic.modules.compressor.Compressor` API).  See DESIGN.md."""
import os as _os

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The branch streams of nn/blocks.py only overlap
# when they land on queues of their own; once RCCL has created its streams (torch.distributed "nccl") the default four
# are shared and the branches serialise (measured: 238 -> 230 images/s on one MI355X).  Read by the HIP runtime when it
# initialises, so this must run before the first device call -- import mcquic_amd (or set the variable) first.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# ROCm 7.2 replays memset NODES of a captured hipGraph wrongly once eager blit work (small device-to-host copies, fills) has run
# between replays: the launch packets it recorded at instantiation point at blit arguments that work reuses.  ATen's two-stage
# reductions zero their semaphores with such a node -- a captured `x.mean()` then returns stale or partial values, no error
# (tools/probes/memset_node_probe.py is the 40-line torch-only reproducer; docs/experiments.md section 9.9).  Nothing this
# package captures holds a memset node, but a caller's loss function or optimizer may: replaying graphs through the ordinary
# command path instead costs nothing measurable here (22.00 -> 22.10 ms per training step, 259.3 -> 259.2 images/s).  Same rule
# as above: read once, when the HIP runtime starts.  `parallel.memset_nodes_replay_correctly()` tells whether it took effect.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

from .modules.compressor import BaseCompressor, Compressor, Neon  # noqa: E402

__all__ = ["BaseCompressor", "Compressor", "Neon"]
__version__ = "0.1.0"
