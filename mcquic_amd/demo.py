"""Single-image compress / restore on the HIP path (reference: mcquic/demo.py:35-134, mcquic/cli.py:40-61).

    python -m mcquic_amd [-qp N] [--local model.mcquic] [--crop] INPUT [OUTPUT]

INPUT an image (.png/.jpg/.jpeg) -> compress to a `.mcq` document; INPUT a `.mcq` document -> restore to .png.
Differences from the reference CLI, all forced by this environment: there is no network, so a model is either a
local checkpoint (`--local`: a torch file with {"config": ..., "model": state_dict} like the released `.mcquic`
files, or a bare state_dict) or, without it, the qp=2 architecture with seeded random weights (plumbing only --
the pictures such a model restores are noise); image I/O is PIL instead of torchvision; there is no CPU mode
(`--disable-gpu` does not exist: the kernels have no CPU fallback).
"""
from __future__ import annotations

import argparse
import pathlib
import sys

import numpy as np
import torch

from .modules.compressor import Compressor
from .utils.specification import File

QP2 = dict(channel=128, m=2, k=[8192, 2048, 512])                     # README.md:304 of the reference


def loadModel(qp: int, local, device) -> Compressor:
    """reference: demo.py:137-163 (download + Config deserialisation replaced by --local / the qp=2 shape)."""
    params = dict(QP2)
    state = None
    if local is not None:
        # weights_only: a checkpoint is tensors + plain containers; never unpickle arbitrary objects from a path that may
        # have come out of a file header
        ckpt = torch.load(str(local), map_location="cpu", weights_only=True)
        state = ckpt.get("model", ckpt) if isinstance(ckpt, dict) else ckpt
        cfg = ckpt.get("config") if isinstance(ckpt, dict) else None
        if isinstance(cfg, dict) and "model" in cfg and "params" in cfg["model"]:
            p = cfg["model"]["params"]
            params = dict(channel=p["channel"], m=p["m"], k=list(p["k"]))
    else:
        torch.manual_seed(3407)
    model = Compressor(**params)
    if state is not None:
        model.load_state_dict(state)
    model.QuantizationParameter = str(local) if local is not None else f"qp_{qp}_msssim"
    return model.to(device).eval()


def detectLocalFile(qp: str):
    """The header's `qp` string names a local checkpoint only if it is an existing file whose suffix contains "mcquic"
    (reference: demo.py:93-97); anything else in that untrusted field is never opened."""
    try:
        path = pathlib.Path(qp)
        if path.exists() and path.is_file() and "mcquic" in path.suffix.lower():
            return path
    except (OSError, ValueError):
        pass
    return None


def compressImage(image: torch.Tensor, model: Compressor, crop: bool) -> File:
    """uint8 [c, h, w] -> File (reference: demo.py:109-122)."""
    image = image.to(torch.float32) / 255.0                            # convert_image_dtype for uint8 inputs
    if crop:
        h, w = image.shape[-2] // 128 * 128, image.shape[-1] // 128 * 128
        top, left = (image.shape[-2] - h) // 2, (image.shape[-1] - w) // 2
        image = image[..., top:top + h, left:left + w]                # AlignedCrop, data/transforms.py:57-78
    image = (image - 0.5) * 2
    _, binaries, headers = model.compress(image[None, ...].contiguous())
    return File(headers[0], binaries[0])


def decompressImage(sourceFile: File, model: Compressor) -> torch.Tensor:
    """File -> uint8 [c, h, w] (reference: demo.py:125-134)."""
    from . import ops
    restored = model.decompress([sourceFile.Content], [sourceFile.FileHeader])
    return ops.detransform(restored[0].contiguous())


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="mcquic_amd", description="Compress / restore a file on MI355X.")
    ap.add_argument("-qp", type=int, default=2, choices=range(0, 14), metavar="[0-13]")
    ap.add_argument("--local", type=pathlib.Path, default=None, help="local model checkpoint")
    ap.add_argument("--crop", action="store_true", help="crop to multiples of 128 instead of padding")
    ap.add_argument("-q", "--quiet", action="store_true")
    ap.add_argument("input", type=pathlib.Path)
    ap.add_argument("output", type=pathlib.Path, nargs="?")
    a = ap.parse_args(argv)
    if not torch.cuda.is_available():
        print("mcquic_amd needs a HIP device (the kernels have no CPU fallback)", file=sys.stderr)
        return 2
    device = torch.device("cuda")
    say = (lambda *s: None) if a.quiet else print
    with torch.inference_mode():
        if a.input.suffix.lower() in (".png", ".jpg", ".jpeg"):
            from PIL import Image
            model = loadModel(a.qp, a.local, device)
            image = torch.from_numpy(np.array(Image.open(a.input).convert("RGB"))).permute(2, 0, 1).to(device)
            target = compressImage(image, model, a.crop)
            raw = a.input.stat().st_size
            say(f"{target.FileHeader.ImageSize} -> {target.size(True)}, {target.BPP:.4f} bpp "
                f"({raw} B => {target.size()} B, compression ratio {(raw - target.size()) / raw * 100:.2f}%)")
            if a.output is not None:
                out = a.output / (a.input.stem + ".mcq") if a.output.is_dir() else a.output
                out.write_bytes(target.serialize())
                say("Saved at", out)
        elif a.input.suffix.lower() == ".mcq":
            from PIL import Image
            source = File.deserialize(a.input.read_bytes())
            local = detectLocalFile(source.FileHeader.QuantizationParameter) or a.local
            if local is None:
                # the reference would now download the model its header names; without a network the restore can only
                # exercise the plumbing -- say so instead of silently decoding with random weights
                print("warning: no checkpoint resolved for this file (pass --local); restoring with the seeded random-weight "
                      "qp=2 model: the output is not the original picture", file=sys.stderr)
            model = loadModel(a.qp, local, device)
            restored = decompressImage(source, model)
            say(f"{source.FileHeader.ImageSize}, {source.BPP:.4f} bpp")
            if a.output is not None:
                out = a.output / (a.input.stem + ".png") if a.output.is_dir() else a.output
                Image.fromarray(restored.permute(1, 2, 0).cpu().numpy()).save(out)
                say("Saved at", out)
        else:
            raise ValueError("Invalid input file.")
    return 0
