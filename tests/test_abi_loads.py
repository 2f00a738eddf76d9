"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/aha_amd.h declares (no compute calls here; compute parity is tests/test_gpu_parity.py)."""
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "aha_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aha_[a-z_0-9]+)\s*\(", text)))


def test_library_is_built_and_exports_every_declared_symbol():
    import ctypes
    from aha_amd import lib
    assert os.path.exists(lib.LIB_PATH), "run __graft_entry__.build() first"
    dll = ctypes.CDLL(lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(dll, name), f"{name} declared in include/aha_amd.h but not exported"
    bound = {n for n, _, _ in lib.SYMBOLS}
    assert bound == set(declared), (bound ^ set(declared))
    assert lib.get().aha_version().startswith(b"aha_amd")


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "aha-_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
    # developer tooling is not allowed to touch the oracle either (its users live under tests/)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tools")):
        for f in files:
            if f.endswith((".py", ".sh")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)
    # bench.py: only inside cpu_baseline(); __graft_entry__.py: only inside smoke()
    import ast
    for f, allowed in (("bench.py", "cpu_baseline"), ("__graft_entry__.py", "smoke")):
        tree = ast.parse(open(os.path.join(ROOT, f)).read())
        for fn in ast.walk(tree):
            if isinstance(fn, (ast.FunctionDef, ast.Module)):
                for node in (fn.body if isinstance(fn, ast.Module) else ast.walk(fn)):
                    names = []
                    if isinstance(node, ast.ImportFrom) and node.module:
                        names = [node.module]
                    elif isinstance(node, ast.Import):
                        names = [a.name for a in node.names]
                    if any(n == "oracle" or n.startswith("oracle.") for n in names):
                        assert isinstance(fn, ast.FunctionDef) and fn.name == allowed, f"{f}: oracle imported outside {allowed}()"


def test_runtime_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import aha_amd
    from aha_amd.runtime import AhaError, Runtime
    with pytest.raises(AhaError):
        Runtime(aha_amd.preset("tiny"), {})
