from .module import (AVAILABLE_GEN_MODULES, Activation, Flatten, GenModule, ParametersGroup,  # noqa: F401
                     SequentialModule, build_gen_module, register_gen_module)
from .linear import FinalLinear, Linear  # noqa: F401
from .conv import Conv2d, FinalConv2d  # noqa: F401
from .norm import BatchNorm1d, BatchNorm2d  # noqa: F401
