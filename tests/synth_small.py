"""Small synthetic genome + TE families + copy table for tests (seeded, pure numpy)."""
import numpy as np

import casegen


def make(seed, n_fam=12, n_chr=3, chr_len=120_000, flank_ok=60, te_type="tir"):
    """te_type shapes the families the way the per-type judges expect them: "tir" terminal inverted repeats + 5-9 bp TSD;
    "helitron" TC...CTRR inserted between A and T, no TSD (judge_Helitron_transposons.py); "non_ltr" a poly-A tail and an
    8-20 bp TSD (judge_Non_LTR_transposons.py)."""
    rng = np.random.default_rng(seed)
    chroms = [list(casegen.rand_seq(rng, chr_len)) for _ in range(n_chr)]
    used = [np.zeros(chr_len, dtype=bool) for _ in range(n_chr)]
    cands, copies, truth, divs = [], [], [], []
    for f in range(n_fam):
        L = int(rng.choice([160, 420, 900, 1500, 2600]))
        ncopy = int(rng.choice([1, 3, 8, 25, 60, 130]))
        div = float(rng.choice([0.02, 0.08, 0.15]))
        tir = casegen.rand_seq(rng, 14)
        if tir.startswith("TG"):
            tir = "CA" + tir[2:]
        cons = tir + casegen.rand_seq(rng, L - 28) + casegen.revcomp(tir)
        tsd_len = int(rng.choice([5, 8, 9]))
        if te_type == "helitron":
            cons = "TC" + cons[2:-4] + str(rng.choice(["CTAG", "CTAA", "CTGG", "CTGA"]))
            tsd_len = 0
        elif te_type == "non_ltr":
            cons = cons[:-16] + "A" * 16
            tsd_len = int(rng.integers(8, 21))
        fam = []
        for k in range(ncopy):
            s = casegen.mutate(rng, cons, div if k else 0.0)
            out = []
            for ch in s[8:-8]:
                x = rng.random()
                if k and x < 0.004:
                    continue
                out.append(ch)
                if k and x > 0.996:
                    out.append(casegen.rand_seq(rng, 1))
            s = s[:8] + "".join(out) + s[-8:]
            minus = bool(rng.integers(0, 2))
            for _try in range(50):
                ci = int(rng.integers(0, n_chr))
                pos = int(rng.integers(flank_ok + 20, chr_len - len(s) - flank_ok - 20))
                if not used[ci][pos - 70:pos + len(s) + 70].any():
                    break
            else:
                continue
            used[ci][pos - 70:pos + len(s) + 70] = True
            tsd = casegen.rand_seq(rng, tsd_len)
            if te_type == "non_ltr":
                while tsd[0] == "A" or tsd[-1] == "A":
                    tsd = casegen.rand_seq(rng, tsd_len)
            ins = casegen.revcomp(s) if minus else s
            chroms[ci][pos:pos + len(s)] = list(ins)
            if te_type == "helitron":      # inserted between A and T (on the element's strand)
                chroms[ci][pos - 1] = "A"
                chroms[ci][pos + len(s)] = "T"
            else:
                # the TSD is a property of the insertion site: the same string on both sides in genome orientation
                chroms[ci][pos - tsd_len:pos] = list(tsd)
                chroms[ci][pos + len(s):pos + len(s) + tsd_len] = list(tsd)
            fam.append((ci, pos + 1, pos + len(s), 1 if minus else 0))
        if not fam:
            continue
        # the candidate = first copy with slightly wrong boundaries, as the coarse stage would hand over
        ci, s1, e1, mn = fam[0]
        dl, dr = int(rng.choice([0, 4, 9])), int(rng.choice([0, 3, 7]))
        if te_type == "helitron" and f % 4:
            dl = dr = 0       # v6 only verifies the 3' end (search_boundary_homo_v4, 'end' side)
        seq = "".join(chroms[ci][s1 - 1 - dl:e1 + dr])
        if mn:
            seq = casegen.revcomp(seq)
        cands.append(seq)
        truth.append(list(fam))
        divs.append(div)
        # a copy finder reports slightly fuzzy ends; keep a few off-contig / too-short entries as well
        cp = []
        for (c2, a, b, m2) in fam:
            cp.append((c2, a - int(rng.integers(0, 3)), b + int(rng.integers(0, 3)), m2))
        if rng.random() < 0.3:
            cp.append((0, 5, 5 + L, 0))  # runs off the contig start -> skipped (Util.py:8103)
        copies.append(cp)
    contigs = ["".join(c) for c in chroms]
    return {"contigs": contigs, "cands": cands, "copies": copies, "truth": truth, "divs": divs}
