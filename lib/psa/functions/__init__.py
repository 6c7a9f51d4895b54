"""Autograd functions of `lib.psa` (the names the reference's lib/psa/functions/__init__.py re-exports)."""
from .psamask import PSAMask, psa_mask

__all__ = ["PSAMask", "psa_mask"]
