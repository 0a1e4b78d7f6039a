"""kiez_amd — MI355X-native exact kNN + hubness reduction behind the `Kiez` API of dobraczka/kiez.

Host Python (this package) mirrors the reference's operator interface; the arithmetic runs in hand-written
HIP kernels (kiez_amd/csrc) reached through the C ABI of include/kiez_amd.h.  There is no CPU fallback."""
from .hubness_reduction import (CSLS, DisSimLocal, HubnessReduction, LocalScaling, MutualProximity,
                                NoHubnessReduction)
from .kiez import Kiez, hubness_reduction_resolver, nn_algorithm_resolver
from .neighbors import NNAlgorithm, NotFittedError, SklearnNN
from . import analysis, evaluate, io  # noqa: F401  (hubness_score, hits, from_openea: "next" rows of SURVEY.md §8 f)

__version__ = "0.1.0"
__all__ = ["Kiez", "NNAlgorithm", "SklearnNN", "HubnessReduction", "NoHubnessReduction", "CSLS", "LocalScaling",
           "MutualProximity", "DisSimLocal", "NotFittedError", "nn_algorithm_resolver", "hubness_reduction_resolver"]
