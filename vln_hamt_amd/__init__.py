"""Importable alias for the product package, which lives in ``vln-hamt_amd/`` (a hyphen is not a
valid Python identifier).  ``import vln_hamt_amd`` resolves every submodule from that directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "vln-hamt_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
