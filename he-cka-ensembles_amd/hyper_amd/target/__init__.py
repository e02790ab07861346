from .mlp import MLPModule  # noqa: F401
from .lenet import LeNet  # noqa: F401
