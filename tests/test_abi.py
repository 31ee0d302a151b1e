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


def test_lib_cluster_host_entry_point():
    """hite_lib_cluster is host code (greedy, sequential by definition): it runs without a GPU; reference goldens + oracle"""
    import ctypes as C
    import numpy as np
    import oracle_lib as O
    from conftest import load_golden

    lib = C.CDLL(os.path.join(ROOT, "hite_amd", "libhite_gpu.so"))
    for c in load_golden("lib_dedup")["chain"]:
        recs = c["recs"]
        n = len(recs)
        col = lambda k, t: np.ascontiguousarray([r[k] for r in recs], dtype=t)  # noqa: E731
        ch, q, qs, qe, s, ss, se = col(0, np.int32), col(1, np.int32), col(2, np.int64), col(3, np.int64), col(4, np.int32), col(5, np.int64), col(6, np.int64)
        sl = np.array(c["lens"], dtype=np.int64)
        cf = np.zeros(n + 3, dtype=np.int64)
        mem = np.zeros(2 * n + 2, dtype=np.int32)
        ncl = C.c_int64(0)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        rc = lib.hite_lib_cluster(C.c_int64(n), p(ch), p(q), p(qs), p(qe), p(s), p(ss), p(se), len(sl), p(sl), C.c_double(c["thr"]),
                                  C.c_int64(n + 2), C.c_int64(2 * n + 2), p(cf), p(mem), C.byref(ncl))
        assert rc == 0
        got = [[int(x) for x in mem[cf[k]:cf[k + 1]]] for k in range(ncl.value)]
        assert got == O.lib_cluster(recs, c["lens"], c["thr"])
        assert [sorted(x) for x in got] == c["clusters"]
