#!/usr/bin/env python
"""Times the reference's OWN Python functions of the hot path in this container beside the C
restatement (the oracle) on the same inputs, so that the C-port figures `bench.py` reports as `cpu_baseline` can be translated
into "reference Python" terms.  TEST / MEASUREMENT INFRASTRUCTURE: imports /root/reference (only possible here) and the oracle.

  FMEA     get_longest_repeats_v4 (Util.py:4122)                 HSP records / s
  judge    remove_sparse_col_in_align_file + judge_boundary_v5    candidates / s   (alignment given: mafft is not timed)
both single process and in an 8-worker process pool (how the reference fans these calls out, Util.py:4775 / :8141).

usage: PYTHONHASHSEED=0 python oracle/time_reference.py [--json out.json]"""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import casegen  # noqa: E402
import ref_harness  # noqa: E402

_U = None


def _util():
    global _U
    if _U is None:
        _U = ref_harness.load_reference_util()
    return _U


def fmea_job(path):
    U = _util()
    pkl = U.get_longest_repeats_v4(path, 2000, 30000, 0)
    return len(U.load_from_file(pkl))


def judge_job(args):
    raw, cand = args
    U = _util()
    clean = U.remove_sparse_col_in_align_file(raw)
    try:
        r = U.judge_boundary_v5(cand, clean, 0, "tir", 1, "cons")
        return bool(r[0])
    except Exception:
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--workers", type=int, default=8)
    a = ap.parse_args()
    import oracle_lib as O

    _util()      # import the reference once, outside the timed regions (forked workers inherit it)
    out = {"host_cores": os.cpu_count(), "workers": a.workers}
    tmp = tempfile.mkdtemp(prefix="hite_time_")
    # ---- FMEA: 8 query files (one per worker), each a blast6 table of one 1 Mbp segment against the others ----------------
    tables, paths, nhsp = [], [], 0
    for k in range(a.workers):
        rows = casegen.make_hsp_table(seed=100 + k, n_seg=4, n_fam=400, noise=1500, frag=(1, 4), copies=(2, 12))
        p = os.path.join(tmp, "q%d.out" % k)
        with open(p, "w") as f:
            f.writelines(casegen.hsp_to_blast6_lines(rows))
        tables.append(rows); paths.append(p); nhsp += len(rows)
    out["fmea_hsp_records"] = nhsp
    out["fmea_input_sha256"] = hashlib.sha256("".join(open(p).read() for p in paths).encode()).hexdigest()
    t0 = time.perf_counter(); n1 = [fmea_job(p) for p in paths]; t1 = time.perf_counter() - t0
    with mp.get_context("fork").Pool(a.workers) as pool:
        t0 = time.perf_counter(); n2 = pool.map(fmea_job, paths); t2 = time.perf_counter() - t0
    assert n1 == n2
    arrs = [O.hsp_arrays([tuple(r) for r in rows]) for rows in tables]   # parsing is not timed on either side of the C call
    t0 = time.perf_counter()
    n3 = [len(O.fmea(h, 2000, 30000)) for h in arrs]
    t3 = time.perf_counter() - t0
    assert n3 == n1, (n3, n1)
    out["fmea"] = {"python_1proc_hsp_per_s": round(nhsp / t1, 1), "python_%dproc_hsp_per_s" % a.workers: round(nhsp / t2, 1),
                   "c_port_1thread_hsp_per_s": round(nhsp / t3, 1), "intervals": int(sum(n1))}
    # ---- judge: synthetic TIR families, 30 copies x ~(te_len + 100) columns, alignment given ----------------------------
    jobs, cases = [], []
    for k in range(64):
        c = casegen.make_msa_case(seed=500 + k, te_type="tir", rows=30, te_len=int(300 + 25 * k), div=0.08, ins_cols=6, trunc_rows=3,
                                  shift_l=(k % 5) - 2, shift_r=(k % 3) - 1, tsd_len=8, tsd_frac=0.85)
        raw = os.path.join(tmp, "aln%d.fa" % k)
        with open(raw, "w") as f:
            for nme, s in zip(c["names"], c["seqs"]):
                f.write(">%s\n%s\n" % (nme, s))
        jobs.append((raw, c["cand"])); cases.append(c)
    out["judge_candidates"] = len(jobs)
    out["judge_cells"] = int(sum(len(c["seqs"]) * len(c["seqs"][0]) for c in cases))
    t0 = time.perf_counter(); r1 = [judge_job(j) for j in jobs]; t1 = time.perf_counter() - t0
    with mp.get_context("fork").Pool(a.workers) as pool:
        t0 = time.perf_counter(); r2 = pool.map(judge_job, jobs); t2 = time.perf_counter() - t0
    assert r1 == r2
    t0 = time.perf_counter()
    r3 = []
    for c in cases:
        m = np.array([list(s.upper().encode()) for s in c["seqs"]], dtype=np.uint8)
        keep = O.sparse_cols(m).astype(bool)
        res = O.judge("tir", np.ascontiguousarray(m[:, keep]), c["cand"], 1)
        r3.append(res[0][0] is True)
    t3 = time.perf_counter() - t0
    assert r3 == r1, (r3, r1)
    out["judge"] = {"python_1proc_cand_per_s": round(len(jobs) / t1, 2), "python_%dproc_cand_per_s" % a.workers: round(len(jobs) / t2, 2),
                    "c_port_1thread_cand_per_s": round(len(jobs) / t3, 1), "is_te": int(sum(r1))}
    print(json.dumps(out, indent=1))
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
