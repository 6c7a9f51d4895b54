"""nn.Module wrappers of `lib.psa` (reference lib/psa/modules/__init__.py)."""
from .psamask import PSAMask

__all__ = ["PSAMask"]
