from .ensemble import (AVAILABLE_GENERATORS, EnsembleParametersGroupGenerator, FixedEnsembleModel,  # noqa: F401
                       build_generator, register_generator)
from .hyper import (GeneratedLayerCodeModelGenerator, LayerCodeModelGenerator, MLPLayerModelGenerator,  # noqa: F401
                    ModelGenerator)
