"""mcquic_amd: MI355X-native encode/decode hot path of McQuic's Compressor (hand-written gfx950 HIP kernels
behind the reference's `mcquic.modules.compressor.Compressor` API).  See DESIGN.md."""
from .modules.compressor import BaseCompressor, Compressor

__all__ = ["BaseCompressor", "Compressor"]
__version__ = "0.1.0"
