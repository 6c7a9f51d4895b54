from .psamask import PSAMask, psa_mask  # noqa: F401
