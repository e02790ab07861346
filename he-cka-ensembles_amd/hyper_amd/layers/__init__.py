from .module import (AVAILABLE_GEN_MODULES, Activation, Flatten, GenModule, ParametersGroup,  # noqa: F401
                     SequentialModule, build_gen_module, register_gen_module)
from .linear import FinalLinear, Linear  # noqa: F401
