"""Import shim: the product package lives in the directory ``aha-_amd/`` (not a valid
Python identifier), so ``import aha_amd`` resolves here and is replaced in
``sys.modules`` by the real package loaded from that directory."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "aha-_amd")
_spec = importlib.util.spec_from_file_location(
    "aha_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["aha_amd"] = _mod
_spec.loader.exec_module(_mod)
