from .specification import CodeSize, FileHeader, ImageSize

__all__ = ["CodeSize", "FileHeader", "ImageSize"]
