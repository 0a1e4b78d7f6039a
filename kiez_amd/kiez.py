"""The `Kiez` facade (mirror of kiez/kiez.py:18-223) over the MI355X exact backend."""
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


class Kiez:
    """Hubness reduced nearest neighbor search for entity alignment — same constructor and methods as the
    reference's `kiez.Kiez` (kiez/kiez.py:98-223); the search and the rescaling run on MI355X.

    >>> from kiez_amd import Kiez
    >>> import numpy as np
    >>> rng = np.random.RandomState(0)
    >>> source, target = rng.rand(100, 50), rng.rand(100, 50)
    >>> k_inst = Kiez(n_candidates=10, algorithm="SklearnNN", hubness="CSLS")
    >>> nn_dist, nn_ind = k_inst.fit(source, target).kneighbors(5)
    """

    def __init__(self, n_candidates: int = 10, algorithm=None, algorithm_kwargs: Optional[Dict[str, Any]] = None,
                 hubness=None, hubness_kwargs: Optional[Dict[str, Any]] = None):
        if not np.issubdtype(type(n_candidates), np.integer):
            raise TypeError(f"n_neighbors does not take {type(n_candidates)} value, enter integer value")
        if n_candidates <= 0:
            raise ValueError(f"Expected n_candidates > 0. Got {n_candidates}")
        if algorithm_kwargs is None:
            algorithm_kwargs = {"n_candidates": n_candidates}
        elif "n_candidates" not in algorithm_kwargs:
            algorithm_kwargs["n_candidates"] = n_candidates
        # the reference tries Faiss first and falls back to SklearnNN (kiez.py:118-122); only the exact
        # backend exists here, so the default resolves to it directly
        algorithm = nn_algorithm_resolver.make(algorithm, algorithm_kwargs)
        assert algorithm
        if hubness_kwargs is None:
            hubness_kwargs = {}
        hubness_kwargs["nn_algo"] = algorithm
        self.hubness = hubness_reduction_resolver.make(hubness, hubness_kwargs)

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
        return (f"Kiez(algorithm: {self.algorithm}, hubness: {self.hubness})"
                f" {self.algorithm._describe_source_target_fitted()}")

    @classmethod
    def from_path(cls, path: Union[str, Path]) -> "Kiez":
        """Load a Kiez instance from a JSON configuration file (kiez.py:154-158)."""
        with open(path) as file:
            return cls(**json.load(file))

    def fit(self, source, target=None) -> "Kiez":
        self.hubness.fit(source, target)
        return self

    def kneighbors_device(self, k: Optional[int] = None):
        """Like `kneighbors` but the (dist, ind) result stays in HBM as DeviceArrays (no PCIe copy)."""
        return self.hubness.kneighbors_device(k)

    def kneighbors(self, k: Optional[int] = None, return_distance: bool = True):
        hubness_reduced_query_dist, query_ind = self.hubness.kneighbors(k)
        if return_distance:
            return hubness_reduced_query_dist, query_ind
        return query_ind
