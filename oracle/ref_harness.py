"""TEST INFRASTRUCTURE ONLY (runs in the build container, never on the GPU box).

Imports the reference's own ``module/Util.py`` from /root/reference with the five
third-party imports that are missing from this image stubbed in ``sys.modules``
(Util.py:19-35): pysam, seaborn, PyPDF2 (unused on the path) and fuzzysearch /
Levenshtein (documented restatements in ``oracle/stubs.py``).

Nothing from the reference is copied: this file only makes ``import Util`` succeed
so that ``oracle/gen_golden.py`` can call the reference's pure-Python functions and
record their inputs/outputs as fixtures under ``tests/golden/``.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("HITE_REFERENCE", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "module", "Util.py"))


def load_reference_util():
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import stubs

    for name in ("pysam", "seaborn", "PyPDF2"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["PyPDF2"].PdfMerger = object
    fz = types.ModuleType("fuzzysearch")
    fz.find_near_matches = stubs.find_near_matches
    sys.modules["fuzzysearch"] = fz
    lv = types.ModuleType("Levenshtein")
    lv.distance = stubs.levenshtein
    sys.modules["Levenshtein"] = lv
    import matplotlib

    matplotlib.use("Agg")
    moddir = os.path.join(REFERENCE_ROOT, "module")
    if moddir not in sys.path:
        sys.path.insert(0, moddir)
    import Util  # noqa: E402  (the reference's module/Util.py)

    return Util


def load_filtr_util():
    """the vendored FiLTR's src/Util.py (bin/FiLTR-main, LTR flank-frame voting); needs the same stubs + intervaltree"""
    import importlib.util

    load_reference_util()
    if "intervaltree" not in sys.modules:
        it = types.ModuleType("intervaltree")
        it.IntervalTree = object
        sys.modules["intervaltree"] = it
    spec = importlib.util.spec_from_file_location("filtr_util", os.path.join(REFERENCE_ROOT, "bin", "FiLTR-main", "src", "Util.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class SyncExecutor:
    """Drop-in for ProcessPoolExecutor that runs submissions inline (deterministic,
    lets monkeypatched callables be used without pickling)."""

    class _F:
        def __init__(self, v):
            self._v = v

        def result(self):
            return self._v

    def __init__(self, *a, **k):
        pass

    def submit(self, fn, *a, **k):
        return SyncExecutor._F(fn(*a, **k))

    def shutdown(self, wait=True):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
