"""A small name -> class resolver with the lookup rules the reference gets from `class_resolver`
(kiez/hubness_reduction/__init__.py:9-12, kiez/neighbors/__init__.py:20-26): names are matched
case-insensitively with the base-class suffix stripped ("NoHubnessReduction" -> "no"); `make` accepts
None (default), a name, a class, or an instance (returned unchanged)."""
from __future__ import annotations

from typing import Any, Dict, Iterable, Optional, Type


def _normalize(name: str, suffix: str) -> str:
    n = name.lower().replace("_", "").replace("-", "").replace(" ", "")
    s = suffix.lower()
    if s and n.endswith(s) and n != s:
        n = n[: -len(s)]
    return n


class Resolver:
    def __init__(self, classes: Iterable[Type], base: Type, default: Optional[Type] = None, suffix: Optional[str] = None,
                 synonyms: Optional[Dict[str, Type]] = None):
        self.base = base
        self.suffix = base.__name__ if suffix is None else suffix
        self.default = default
        self.lookup_dict: Dict[str, Type] = {_normalize(c.__name__, self.suffix): c for c in classes}
        self.synonyms = {_normalize(k, self.suffix): v for k, v in (synonyms or {}).items()}

    @property
    def options(self):
        return set(self.lookup_dict)

    def lookup(self, query) -> Type:
        if query is None:
            if self.default is None:
                raise ValueError("no default given")
            return self.default
        if isinstance(query, str):
            key = _normalize(query, self.suffix)
            if key in self.lookup_dict:
                return self.lookup_dict[key]
            if key in self.synonyms:
                return self.synonyms[key]
            raise KeyError(f"Invalid query: {query}. Try one of: {sorted(self.options)}")
        if isinstance(query, type) and issubclass(query, self.base):
            return query
        raise TypeError(f"Invalid query type: {type(query)} ({query!r})")

    def make(self, query, pos_kwargs: Optional[Dict[str, Any]] = None, **kwargs):
        if query is not None and not isinstance(query, (str, type)):
            if isinstance(query, self.base):
                return query  # an instance is passed through unchanged
            raise TypeError(f"Invalid query type: {type(query)} ({query!r})")
        cls = self.lookup(query)
        return cls(**(pos_kwargs or {}), **kwargs)
