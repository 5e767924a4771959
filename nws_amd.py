"""Importable alias of the package directory `neural-waveshaping-synthesis_amd` (hyphens cannot be
written in an import statement):  ``import nws_amd as nws``."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("neural-waveshaping-synthesis_amd")
sys.modules[__name__] = _pkg
# `from nws_amd.<sub> import x` must resolve to the SAME module objects (a second copy of _lib would re-declare the ctypes
# structures and prototypes on the shared library handle)
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_pkg.__name__ + "."):
        sys.modules[__name__ + _name[len(_pkg.__name__):]] = _mod
