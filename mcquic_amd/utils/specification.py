"""Boundary types returned by `Compressor.compress` and the `.mcq` container
(reference: mcquic/utils/specification.py:56-183, mcquic/utils/__init__.py:32-48).

Plain dataclasses with the reference's field / property names.  `File.serialize` writes the same msgpack document the
reference's marshmallow `FileSchema().dump()` + `msgpack.packb(use_bin_type=True)` produce:
    {"fileHeader": {"qp", "version", "codeSize": {"m", "heights", "widths", "k"}, "imageSize": {"height", "width", "channel"}},
     "contents": [bytes, ...]}
(marshmallow is not installed here, so the reference's serializer cannot be run: the layout is restated from the
schema declarations, specification.py:22-53, and pinned byte for byte by a document derived by hand from those
declarations and the msgpack format, tests/test_mcq_container.py; a cross-read of files written by the reference itself
stays unexecuted.)
"""
from __future__ import annotations

import warnings
from dataclasses import dataclass
from typing import List, Union

__all__ = ["ImageSize", "CodeSize", "FileHeader", "File", "versionCheck"]

VERSION = "0.1.40"        # the reference snapshot's mcquic.__version__


def _parse(version: str):
    parts = version.split(".")
    if len(parts) != 3 or not all(p.isdigit() for p in parts):
        raise ValueError(f"invalid version number '{version}'")
    return tuple(int(p) for p in parts)


def versionCheck(versionStr: str) -> bool:
    """Same rule as mcquic/utils/__init__.py:32-48: newer or major-different files are rejected, minor mismatch warns."""
    version, builtIn = _parse(versionStr), _parse(VERSION)
    if builtIn < version:
        raise ValueError(f"Version too new. Given {versionStr}, but I'm {VERSION} now.")
    if version[0] != builtIn[0]:
        raise ValueError(f"Major version mismatch. Given {versionStr}, but I'm {VERSION} now.")
    if version[1] != builtIn[1]:
        warnings.warn(f"Minor version mismatch. Given {versionStr}, but I'm {VERSION} now.")
    return True


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

    def __str__(self) -> str:
        sequence = ", ".join(f"[{w}x{h}, {k}]x{m}" for h, w, k, m in zip(self.heights, self.widths, self.k, self.m))
        return f"\n        {self.m} code-groups: {sequence}"


@dataclass(init=False)
class FileHeader:
    qp: str
    version: str
    codeSize: CodeSize
    imageSize: ImageSize

    def __init__(self, version: str, qp: str, codeSize: CodeSize, imageSize: ImageSize):
        versionCheck(version)                      # raises for files this build cannot read
        self.qp, self.version, self.codeSize, self.imageSize = qp, version, codeSize, imageSize

    # the reference's read-only aliases of the four fields
    QuantizationParameter = property(lambda self: str(self.qp))
    Version = property(lambda self: self.version)
    CodeSize = property(lambda self: self.codeSize)
    ImageSize = property(lambda self: self.imageSize)


@dataclass
class File:
    fileHeader: FileHeader
    contents: List[bytes]

    FileHeader = property(lambda self: self.fileHeader)
    Content = property(lambda self: self.contents)

    def serialize(self) -> bytes:
        import msgpack
        h = self.fileHeader
        doc = {"fileHeader": {"qp": str(h.qp), "version": h.version,
                              "codeSize": {"m": list(h.codeSize.m), "heights": list(h.codeSize.heights),
                                           "widths": list(h.codeSize.widths), "k": list(h.codeSize.k)},
                              "imageSize": {"height": h.imageSize.height, "width": h.imageSize.width, "channel": h.imageSize.channel}},
               "contents": list(self.contents)}
        return msgpack.packb(doc, use_bin_type=True)

    @staticmethod
    def deserialize(data: bytes) -> "File":
        import msgpack
        doc = msgpack.unpackb(data, use_list=False, raw=False)
        try:
            h = doc["fileHeader"]
            cs, im = h["codeSize"], h["imageSize"]
            contents = list(doc["contents"])
            if not contents or not all(isinstance(c, bytes) and c for c in contents):
                raise ValueError("Invalid value")
            header = FileHeader(h["version"], h["qp"], CodeSize(list(cs["m"]), list(cs["heights"]), list(cs["widths"]), list(cs["k"])),
                                ImageSize(int(im["height"]), int(im["width"]), int(im["channel"])))
        except (KeyError, TypeError) as e:
            raise ValueError(f"not a .mcq document: {e}") from e
        return File(header, contents)

    BPP = property(lambda self: self.size() * 8 / self.fileHeader.imageSize.Pixels)      # bits per image pixel

    def size(self, human: bool = False) -> Union[int, str]:
        size = sum(len(x) for x in self.contents)
        if not human:
            return size
        for unit in ("B", "KiB", "MiB", "GiB"):
            if size < 1024 or unit == "GiB":
                return f"{size:.2f}{unit}" if unit != "B" else f"{size}B"
            size /= 1024
