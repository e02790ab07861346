from .ensemble import (AVAILABLE_GENERATORS, EnsembleParametersGroupGenerator, FixedEnsembleModel,  # noqa: F401
                       build_generator, register_generator)
