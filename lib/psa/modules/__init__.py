from .psamask import PSAMask  # noqa: F401
