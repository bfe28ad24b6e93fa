"""Import alias: ``import sharkshark4k_amd`` loads the package in ``sharkshark-4k_amd/``.

The package directory keeps the name the project layout prescribes (it contains a hyphen, which
Python cannot import directly); this shim registers it in ``sys.modules`` under an importable name.
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "sharkshark-4k_amd")
_spec = _ilu.spec_from_file_location("sharkshark4k_amd", _os.path.join(_dir, "__init__.py"),
                                     submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["sharkshark4k_amd"] = _mod
_spec.loader.exec_module(_mod)
