"""Load the hot-path modules of the reference (dobraczka/kiez @ /root/reference) by file path.

Only usable in the build container (the reference tree does not travel to the GPU box).
`import kiez` fails there (class_resolver is not installed), so the package
``__init__`` files are bypassed: empty package objects are seeded into ``sys.modules`` and the
individual hot-path files are executed with ``spec_from_file_location``.

Nothing from the reference is copied: this module only *imports* it to produce golden vectors.
"""
import importlib.util
import sys
import types
from pathlib import Path

REF = Path("/root/reference")


def _pkg(name):
    m = types.ModuleType(name)
    m.__path__ = []  # mark as package
    sys.modules[name] = m
    return m


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, REF / relpath)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    """Return a namespace with SklearnNN, NoHubnessReduction, CSLS, MutualProximity, LocalScaling, DisSimLocal."""
    if not REF.exists():
        raise RuntimeError("reference tree not present (only available in the build container)")
    kiez = _pkg("kiez")
    neighbors = _pkg("kiez.neighbors")
    _pkg("kiez.neighbors.exact")
    _pkg("kiez.hubness_reduction")
    kiez.neighbors = neighbors
    base = _load("kiez.neighbors.neighbor_algorithm_base", "kiez/neighbors/neighbor_algorithm_base.py")
    neighbors.NNAlgorithm = base.NNAlgorithm
    sk = _load("kiez.neighbors.exact.sklearn_nearest_neighbors", "kiez/neighbors/exact/sklearn_nearest_neighbors.py")
    hb = _load("kiez.hubness_reduction.base", "kiez/hubness_reduction/base.py")
    csls = _load("kiez.hubness_reduction.csls", "kiez/hubness_reduction/csls.py")
    mp = _load("kiez.hubness_reduction.mutual_proximity", "kiez/hubness_reduction/mutual_proximity.py")
    ls = _load("kiez.hubness_reduction.local_scaling", "kiez/hubness_reduction/local_scaling.py")
    dsl = _load("kiez.hubness_reduction.dis_sim", "kiez/hubness_reduction/dis_sim.py")
    ns = types.SimpleNamespace(
        NNAlgorithm=base.NNAlgorithm,
        SklearnNN=sk.SklearnNN,
        HubnessReduction=hb.HubnessReduction,
        NoHubnessReduction=hb.NoHubnessReduction,
        CSLS=csls.CSLS,
        MutualProximity=mp.MutualProximity,
        LocalScaling=ls.LocalScaling,
        DisSimLocal=dsl.DisSimLocal,
    )
    return ns
