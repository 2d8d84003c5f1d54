"""Importable alias of the package directory `vae-posterior-consistency_amd/` (its name has a hyphen).

    import vpc_amd as vpc
    model = vpc.Reg_VAE(...)
"""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("vae-posterior-consistency_amd")
for _name, _mod in list(sys.modules.items()):
    if _name.startswith("vae-posterior-consistency_amd."):
        sys.modules["vpc_amd." + _name.split(".", 1)[1]] = _mod
sys.modules[__name__] = _pkg
