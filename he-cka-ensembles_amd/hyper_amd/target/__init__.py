from .mlp import MLPModule  # noqa: F401
