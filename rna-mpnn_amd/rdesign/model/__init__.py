from .rdesign import RNAModel  # noqa: F401
