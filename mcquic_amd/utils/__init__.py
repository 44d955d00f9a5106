from .specification import CodeSize, File, FileHeader, ImageSize, versionCheck

__all__ = ["CodeSize", "File", "FileHeader", "ImageSize", "versionCheck"]
