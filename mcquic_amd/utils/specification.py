"""Boundary types returned by `Compressor.compress` (reference: mcquic/utils/specification.py:56-156).

Plain dataclasses with the reference's field and property names.  The msgpack/marshmallow `.mcq`
container itself is a "next" row and is not implemented here.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

__all__ = ["ImageSize", "CodeSize", "FileHeader"]


@dataclass
class ImageSize:
    height: int
    width: int
    channel: int

    @property
    def Pixels(self) -> int:
        return self.height * self.width

    def __str__(self) -> str:
        return f"[{self.width}x{self.height}, {self.channel}]"


@dataclass
class CodeSize:
    """m: groups per level, heights / widths: latent size per level, k: codewords per level."""
    m: List[int]
    heights: List[int]
    widths: List[int]
    k: List[int]


@dataclass
class FileHeader:
    version: str
    qp: str
    codeSize: CodeSize
    imageSize: ImageSize

    @property
    def QuantizationParameter(self) -> str:
        return str(self.qp)

    @property
    def Version(self) -> str:
        return self.version

    @property
    def CodeSize(self) -> CodeSize:
        return self.codeSize

    @property
    def ImageSize(self) -> ImageSize:
        return self.imageSize
