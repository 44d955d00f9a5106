from .compressor import BaseCompressor, Compressor

__all__ = ["BaseCompressor", "Compressor"]
