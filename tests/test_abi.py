"""CPU-only: the C-ABI library builds, loads and exports every symbol include/hite_gpu.h declares,
and the product fails loudly (no fallback) when there is no GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "hite_gpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hite_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    import __graft_entry__ as g

    g.build()
    lib = ctypes.CDLL(os.path.join(ROOT, "hite_amd", "libhite_gpu.so"))
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), "missing export: " + s
    assert lib.hite_version() >= 1


def test_no_cpu_fallback_without_gpu():
    import torch

    import hite_amd

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hite_amd.HiteError):
        hite_amd.Context(0)


def test_product_does_not_import_oracle():
    """only tests/, smoke() and bench.py's cpu_baseline leg may import / link / load the oracle"""
    bad = re.compile(r"^\s*(import|from)\s+\S*oracle|#\s*include\s+\S*oracle|oracle_lib|libhite_oracle|dlopen|CDLL\(.*oracle", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "hite_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not bad.search(txt), f
    mk = open(os.path.join(ROOT, "hite_amd", "csrc", "Makefile")).read()
    assert "oracle" not in mk
