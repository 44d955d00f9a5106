from .compressor import BaseCompressor, Compressor, Neon

__all__ = ["BaseCompressor", "Compressor", "Neon"]
