"""The `Kiez` facade over the MI355X exact backend.

Drop-in for the reference's `kiez.Kiez` (kiez/kiez.py:18-223): same constructor arguments, methods, properties, error
types and message texts; everything between `fit` and the returned `(dist, ind)` runs on the GPU through the C ABI.
The facade itself is three small pieces: argument validation (`_candidate_count`), the two name resolvers, and thin
delegation to the resolved `HubnessReduction` object, which owns the `NNAlgorithm`.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Any, Dict, List, Optional, Union

import numpy as np

from .hubness_reduction import (CSLS, DisSimLocal, HubnessReduction, LocalScaling, MutualProximity,
                                NoHubnessReduction)
from .neighbors import NNAlgorithm, SklearnNN, available_nn_algorithms
from .resolver import Resolver

# kiez/neighbors/__init__.py:20-26 (default SklearnNN) and kiez/hubness_reduction/__init__.py:9-12 (default "no")
nn_algorithm_resolver = Resolver([SklearnNN], base=NNAlgorithm, default=SklearnNN,
                                 synonyms={"exact": SklearnNN, "hip": SklearnNN, "mi355x": SklearnNN})
hubness_reduction_resolver = Resolver([NoHubnessReduction, CSLS, LocalScaling, MutualProximity, DisSimLocal],
                                      base=HubnessReduction, default=NoHubnessReduction)


def _candidate_count(value) -> int:
    """The reference's checks on `n_candidates` (kiez.py:106-113): integer type first, then positivity."""
    if not np.issubdtype(type(value), np.integer):
        raise TypeError(f"n_neighbors does not take {type(value)} value, enter integer value")
    if value <= 0:
        raise ValueError(f"Expected n_candidates > 0. Got {value}")
    return value


class Kiez:
    """Hubness reduced nearest neighbor search for entity alignment on MI355X.

    Parameters are the reference's: `n_candidates` (candidates fetched per query before hubness reduction), `algorithm`
    / `algorithm_kwargs` (an `NNAlgorithm` instance, class or name; only the exact backend exists here and it is the
    default), `hubness` / `hubness_kwargs` (a `HubnessReduction` instance, class or name; default: none).

    >>> from kiez_amd import Kiez
    >>> import numpy as np
    >>> rng = np.random.RandomState(0)
    >>> source, target = rng.rand(100, 50), rng.rand(100, 50)
    >>> k_inst = Kiez(n_candidates=10, algorithm="SklearnNN", hubness="CSLS")
    >>> nn_dist, nn_ind = k_inst.fit(source, target).kneighbors(5)
    """

    def __init__(self, n_candidates: int = 10, algorithm=None, algorithm_kwargs: Optional[Dict[str, Any]] = None,
                 hubness=None, hubness_kwargs: Optional[Dict[str, Any]] = None):
        n_candidates = _candidate_count(n_candidates)
        # the NN backend learns the candidate count unless its own kwargs already name one (kiez.py:114-117)
        nn_kwargs = {"n_candidates": n_candidates} if algorithm_kwargs is None else algorithm_kwargs
        nn_kwargs.setdefault("n_candidates", n_candidates)
        # (the reference tries Faiss first and falls back to SklearnNN, kiez.py:118-122; the default here IS the exact one)
        nn_algo = nn_algorithm_resolver.make(algorithm, nn_kwargs)
        assert nn_algo
        hub_kwargs = {} if hubness_kwargs is None else hubness_kwargs
        hub_kwargs["nn_algo"] = nn_algo
        self.hubness = hubness_reduction_resolver.make(hubness, hub_kwargs)

    # ---- introspection ------------------------------------------------------------------------------
    @staticmethod
    def show_algorithm_options() -> List[str]:
        return available_nn_algorithms(as_string=True)

    @staticmethod
    def show_hubness_options() -> List[str]:
        return list(hubness_reduction_resolver.options)

    @property
    def algorithm(self):
        return self.hubness.nn_algo

    @algorithm.setter
    def algorithm(self, value):
        self.hubness.nn_algo = value

    def __repr__(self):
        fitted = self.algorithm._describe_source_target_fitted()
        return f"Kiez(algorithm: {self.algorithm}, hubness: {self.hubness}) {fitted}"

    @classmethod
    def from_path(cls, path: Union[str, Path]) -> "Kiez":
        """A Kiez instance from a JSON file of constructor arguments (kiez.py:154-158)."""
        return cls(**json.loads(Path(path).read_text()))

    # ---- the hot path -------------------------------------------------------------------------------
    def fit(self, source, target=None) -> "Kiez":
        self.hubness.fit(source, target)
        return self

    def kneighbors_device(self, k: Optional[int] = None):
        """Like `kneighbors` but the (dist, ind) result stays in HBM as DeviceArrays (no PCIe copy)."""
        return self.hubness.kneighbors_device(k)

    def kneighbors(self, k: Optional[int] = None, return_distance: bool = True):
        dist, ind = self.hubness.kneighbors(k)
        return (dist, ind) if return_distance else ind
