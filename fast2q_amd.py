"""Alias so that ``import fast2q_amd`` works (the package directory is ``2fast2q_amd``)."""
import importlib as _il
import sys as _sys

_pkg = _il.import_module("2fast2q_amd")
_sys.modules[__name__] = _pkg
