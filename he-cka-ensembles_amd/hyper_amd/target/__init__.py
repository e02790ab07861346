from .mlp import MLPModule  # noqa: F401
from .lenet import LeNet  # noqa: F401
from .resnet import ResNet, resnet18  # noqa: F401
