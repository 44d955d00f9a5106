"""mcquic_amd: MI355X-native encode/decode hot path of McQuic's Compressor (hand-written gfx950 HIP kernels
behind the reference's `mcquic.modules.compressor.Compressor` API).  See DESIGN.md."""
import os as _os

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The branch streams of nn/blocks.py only overlap
# when they land on queues of their own; once RCCL has created its streams (torch.distributed "nccl") the default four
# are shared and the branches serialise (measured: 238 -> 230 images/s on one MI355X).  Read by the HIP runtime when it
# initialises, so this must run before the first device call -- import mcquic_amd (or set the variable) first.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .modules.compressor import BaseCompressor, Compressor, Neon  # noqa: E402

__all__ = ["BaseCompressor", "Compressor", "Neon"]
__version__ = "0.1.0"
