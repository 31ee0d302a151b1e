"""Tandem-repeat masking: the build's own masker where the reference runs TRF (Util.py:2855-2874).
PARITY UNPINNED (TRF is third-party): the masker is MEASURED -- against the planted arrays, and against the masks TRF 4.09
itself (the binary bundled with the reference, run by oracle/gen_golden.py with the reference's command line) produced on the
same sequences (tests/golden/trf_mask.json.gz).  CPU: the twin (oracle/hite_oracle_trf.c).  GPU: HIP == twin bit for bit."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import casegen  # noqa: E402
import oracle_lib as O  # noqa: E402
from conftest import load_golden  # noqa: E402


def twin_mask(contigs, max_period=500):
    seqs = [c.encode() if isinstance(c, str) else bytes(c) for c in contigs]
    buf = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    out = np.zeros(len(buf), dtype=np.uint8)
    L = O.lib()
    L.orc_tr_mask.restype = C.c_int64
    n = L.orc_tr_mask(buf.ctypes.data_as(O.u8p), off.ctypes.data_as(O.i64p), len(seqs), max_period, out.ctypes.data_as(O.u8p))
    assert n == int(out.sum())
    return out.astype(bool)


def _measure(mask, case):
    G = len(case["seq"])
    planted = np.zeros(G, dtype=bool)
    clean = np.zeros(G, dtype=bool)       # arrays TRF's scheme is meant to find: <= 8 % substitutions per copy, >= 2 copies
    for a, b, p, copies, div, ind in case["planted"]:
        planted[a:b] = True
        if div <= 0.08 and copies >= 2.0 and (b - a) >= 30:
            clean[a:b] = True
    trf = np.zeros(G, dtype=bool)
    for a, b in case["trf_masked"]:
        trf[a:b] = True
    return dict(planted_recall=(mask & planted).sum() / planted.sum(), clean_recall=(mask & clean).sum() / clean.sum(),
                outside=int((mask & ~planted).sum()), precision_vs_planted=(mask & planted).sum() / max(1, mask.sum()),
                recall_vs_trf=(mask & trf).sum() / trf.sum(), trf_planted_recall=(trf & planted).sum() / planted.sum(),
                trf_clean_recall=(trf & clean).sum() / clean.sum())


def test_twin_against_planted_arrays_and_trf():
    tot = []
    for case in load_golden("trf_mask"):
        seq, planted = casegen.make_tandem_case(case["seed"])
        assert seq == case["seq"] and planted == case["planted"]          # the fixture's sequences are the generator's
        m = _measure(twin_mask([case["seq"]]), case)
        tot.append(m)
        print("seed %d: %s" % (case["seed"], {k: round(float(v), 3) for k, v in m.items()}))
    avg = {k: float(np.mean([t[k] for t in tot])) for k in tot[0]}
    # measured (3 x 120 kb, ~200 arrays of period 1-500, up to 15 % substitutions and 2 % indels per copy), edit penalty 5 (round 4;
    # penalty 7 in brackets): planted bases masked 0.944 [0.80] (TRF itself: 0.936), arrays with <= 8 % substitutions 0.9985 [0.96]
    # (TRF 0.985), 59-93 [40-45] bases masked outside any planted array per sequence (TRF 30-37; precision 0.998), 0.988 [0.85] of
    # TRF's own mask covered.  The masker compares a copy with its neighbour (twice the divergence TRF's consensus sees), hence the
    # lower penalty; oracle/hite_oracle_trf.c, "calibration".
    assert avg["planted_recall"] >= 0.92 and avg["clean_recall"] >= 0.99
    assert avg["precision_vs_planted"] >= 0.995 and max(t["outside"] for t in tot) < 120
    assert avg["recall_vs_trf"] >= 0.97


def test_twin_leaves_dispersed_repeats_alone():
    """copies of one element far apart (what the pipeline is looking for) are not tandem repeats; contig borders are respected"""
    rng = np.random.default_rng(5)
    te = casegen.rand_seq(rng, 900)
    a = casegen.rand_seq(rng, 3000) + te + casegen.rand_seq(rng, 4000) + casegen.mutate(rng, te, 0.05) + casegen.rand_seq(rng, 2000)
    b = "ACGTTGCA" * 4 + casegen.rand_seq(rng, 500)         # a 32-base array at the contig start, the same unit ends contig a
    a = a + "ACGTTGCA" * 2
    m = twin_mask([a, b])
    assert m[:len(a) - 16].sum() == 0                       # the dispersed copies stay, 16 bases at the end of a are too few
    assert m[len(a) - 16:len(a)].sum() == 0                 # ... and do not join the array across the contig border
    assert m[len(a):len(a) + 32].all() and m[len(a) + 48:].sum() == 0


@pytest.mark.gpu
def test_gpu_masker_equals_twin(tmp_path):
    import hite_amd
    from hite_amd import util

    ctx = hite_amd.Context(0)
    try:
        for case in load_golden("trf_mask")[:2]:
            # several contigs, an N run, a short last contig
            s = case["seq"]
            contigs = [s[:50_000], s[50_000:50_777] + "N" * 40 + s[50_777:90_000], s[90_000:], "ACGT" * 10]
            ctx.genome_pack(contigs)
            got = ctx.tr_mask(500)
            exp = twin_mask(contigs)
            assert got.shape == exp.shape and np.array_equal(got, exp), int((got != exp).sum())
            assert exp.sum() > 10_000
        # the host-side mirror: filter_tandem_repeats without `trf` on the PATH masks the chunk with the GPU masker
        util._CTX = ctx
        names = ["chr1$0", "chr1$1000000"]
        seqs = {names[0]: load_golden("trf_mask")[2]["seq"][:60_000], names[1]: load_golden("trf_mask")[2]["seq"][60_000:]}
        os.environ["HITE_TR_MASKER"] = "gpu"
        try:
            out = util.filter_tandem_repeats(names, seqs, str(tmp_path), 0, 1)
        finally:
            del os.environ["HITE_TR_MASKER"]
        n2, c2 = util.read_fasta(out)
        assert n2 == names
        exp = twin_mask([seqs[n] for n in names])
        got = np.concatenate([np.frombuffer(c2[n].encode(), dtype=np.uint8) == ord("N") for n in names])
        assert np.array_equal(got, exp)
    finally:
        util._CTX = None
        ctx.close()
