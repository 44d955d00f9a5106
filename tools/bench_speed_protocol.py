#!/usr/bin/env python3
"""The reference's own throughput protocol (mcquic/validate/validator.py:60-97 = mcquic_amd.validate.speed): random
10x3x768x512 batch, 50 compress + 50 decompress calls (byte streams included), Mpps each way."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mcquic_amd import Compressor, validate  # noqa: E402

torch.manual_seed(3407)
model = Compressor(128, 2, [8192, 2048, 512]).eval().to("cuda:0")
enc, dec = validate.speed(model)
print(f"speed protocol (batch 10, 768x512, compress/decompress incl. rANS): encode {enc:.1f} Mpps, decode {dec:.1f} Mpps")
