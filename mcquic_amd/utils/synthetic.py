"""The synthetic workload of BASELINE.json's configs[1] / configs[2] as bench.py generates it -- one place, so that the bench,
the parity tools, the golden-vector capture (tests/golden/make_golden.py, F5c) and the GPU tests all see the same bytes:

    weights   torch.manual_seed(3407); Compressor(128, 2, [8192, 2048, 512])      (same random init on every rank)
    images    rank r of `bench.py --gpus N`: 2 * U[0, 1) - 1 from torch.Generator("cpu").manual_seed(3407 + r)

configs[2] (256 images sharded 8-way) is shards r = 0 .. 7 of 32 images each; configs[1] is shard 0.
Seed 3407 is the reference's (mcquic/train/utils.py:332)."""
from __future__ import annotations

import hashlib
from typing import Dict

import torch

SEED = 3407
QP2_MODEL = dict(channel=128, m=2, k=[8192, 2048, 512])
HEIGHT, WIDTH = 768, 512


def bench_model(seed: int = SEED):
    """The qp=2 model with bench.py's random-init weights (CPU, eval mode)."""
    from ..modules.compressor import Compressor
    torch.manual_seed(seed)
    return Compressor(**QP2_MODEL).eval()


def bench_state_dict(seed: int = SEED) -> Dict[str, torch.Tensor]:
    return {k: v.detach().clone() for k, v in bench_model(seed).state_dict().items()}


def bench_images(rank: int, n: int = 32, height: int = HEIGHT, width: int = WIDTH, seed: int = SEED) -> torch.Tensor:
    """fp32 [n, 3, height, width] in [-1, 1): the batch rank `rank` encodes."""
    g = torch.Generator(device="cpu").manual_seed(seed + rank)
    return torch.rand((n, 3, height, width), generator=g) * 2 - 1


def code_hash(code_img: torch.Tensor) -> bytes:
    """First 8 bytes of SHA-256 over one image's codes of one level as little-endian int16 [m, h, w]."""
    return hashlib.sha256(code_img.detach().cpu().contiguous().numpy().astype("<i2").tobytes()).digest()[:8]


def state_dict_sha(sd: Dict[str, torch.Tensor]) -> str:
    return hashlib.sha256(b"".join(sd[k].detach().cpu().contiguous().numpy().tobytes() for k in sorted(sd))).hexdigest()


# ---- BASELINE configs[3]: the VQ distance / argmin kernel in isolation ------------------------------------
#   latents x ~ N(0, 0.1^2) [32, m * d, 48, 32], codebook ~ N(0, 2 / (5 d)) [m, k, d] (the init law of mcquic/modules/quantizer.py:398),
#   both from ONE torch.Generator seeded 0, latents drawn first -- the tensors bench.py's `vq_config4` leg times, the golden capture
#   F2b (tests/golden/make_golden.py f2b) runs the reference on, and tests/test_gpu_fullsize.py checks every code of.
VQ_CASES = {"config4": dict(m=4, k=4096, d=256, n=32, h=48, w=32, seed=0),       # BASELINE.json configs[3], 49 152 vectors per codebook
            "qp2_l0": dict(m=2, k=8192, d=64, n=32, h=48, w=32, seed=1)}        # the qp=2 model's level 0 at configs[1]'s batch


def vq_case(tag: str):
    """(latents [n, m * d, h, w], codebook [m, k, d]) of VQ_CASES[tag] on the CPU."""
    c = VQ_CASES[tag]
    g = torch.Generator(device="cpu").manual_seed(c["seed"])
    lat = torch.randn((c["n"], c["m"] * c["d"], c["h"], c["w"]), generator=g) * 0.1
    cb = torch.randn((c["m"], c["k"], c["d"]), generator=g) * (2 / (5 * c["d"])) ** 0.5
    return lat, cb
