"""HIP-backed counterparts of the mcquic.nn layers that sit on the Compressor encode/decode path."""
from .convs import Conv2d, PixelShuffle3x3, conv1x1, conv3x3, pixelShuffle3x3
from .gdn import GenDivNorm, InvGenDivNorm
from .blocks import AttentionBlock, GroupNorm, ResidualBlock, ResidualBlockShuffle, ResidualBlockWithStride

__all__ = ["Conv2d", "PixelShuffle3x3", "conv1x1", "conv3x3", "pixelShuffle3x3", "GenDivNorm", "InvGenDivNorm",
           "AttentionBlock", "GroupNorm", "ResidualBlock", "ResidualBlockShuffle", "ResidualBlockWithStride"]
