"""Importable alias of the package directory `vae-cyclegan-implementation_amd` (whose name is
not a valid Python identifier): `import vcg_amd; vcg_amd.Networks.CycleVAEGAN(...)`."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("vae-cyclegan-implementation_amd")
sys.modules[__name__] = _pkg
