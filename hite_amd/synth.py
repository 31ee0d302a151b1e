"""Synthetic workload generator for bench.py and the full-size tests: a random
genome (GC 0.42, 50 Mbp chromosomes) with TIR and LTR families planted as diverged, partly
truncated copies with target-site duplications, plus the candidate file and the copy table that
the fine stage consumes.  Bench/test tooling only -- not part of the product path.

Everything random derives from numpy PCG64 / torch Philox seeds, so a (config, seed) pair names
the workload.  The background genome is generated on the GPU when one is given, the planted
copies with numpy on the host.
"""
import numpy as np

ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.array([3, 2, 1, 0], dtype=np.uint8)  # on codes


def _rand_codes(rng, n, gc=0.42):
    p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    return rng.choice(4, size=n, p=p).astype(np.uint8)


def _mutate_copy(rng, cons, div, indel_rate):
    s = cons.copy()
    L = len(s)
    m = rng.random(L) < div
    k = int(m.sum())
    if k:
        s[m] = (s[m] + rng.integers(1, 4, size=k).astype(np.uint8)) & 3
    n_ind = rng.binomial(max(L - 40, 1), indel_rate)
    if n_ind:
        pos = np.sort(rng.integers(20, L - 20, size=n_ind))
        is_del = rng.random(n_ind) < 0.5
        dpos = pos[is_del]
        ipos = pos[~is_del]
        if len(ipos):
            s = np.insert(s, ipos, rng.integers(0, 4, size=len(ipos)).astype(np.uint8))
            # deletion positions shift by the insertions before them
            dpos = dpos + np.searchsorted(ipos, dpos, side="right")
        if len(dpos):
            s = np.delete(s, np.unique(dpos))
    return s


def make_families(rng, n_tir, n_ltr, max_copies=300):
    fams = []
    for f in range(n_tir + n_ltr):
        is_ltr = f >= n_tir
        if not is_ltr:
            L = int(rng.integers(150, 3001))
            tl = int(rng.integers(10, 41))
            tir = _rand_codes(rng, tl)
            if tir[0] == 3 and tir[1] == 2:  # avoid TG...CA
                tir[0] = 1
            body = _rand_codes(rng, max(L - 2 * tl, 20))
            right = COMP[tir[::-1]].copy()
            dv = rng.random(tl) < 0.05  # TIR arms at <= 10 % divergence
            right[dv] = (right[dv] + 1) & 3
            cons = np.concatenate([tir, body, right])
            tsd = int(rng.choice([2, 3, 4, 5, 6, 8, 9, 10, 11]))
        else:
            ltr = _rand_codes(rng, int(rng.integers(100, 1501)))
            ltr[0], ltr[1], ltr[-2], ltr[-1] = 3, 2, 1, 0  # TG ... CA
            internal = _rand_codes(rng, int(rng.integers(1000, 8001)))
            cons = np.concatenate([ltr, internal, ltr])
            tsd = 5
        ncopy = int(min(max_copies, 2 + rng.geometric(0.05)))
        fams.append(dict(cons=cons, tsd=tsd, ncopy=ncopy, ltr=is_ltr))
    return fams


def make_workload(genome_bp=100_000_000, n_tir=500, n_ltr=0, cands_per_family=10, seed=20250927, chrom_bp=50_000_000,
                  device=None, flank=50, cand_seed=None, family_seed=None, family_keep=1.0):
    """-> dict(genome (uint8 ASCII, torch tensor on `device` or numpy), contig_off, cands (uint8 array),
    cand_off, copy_first, contig, start1, end1, minus, family, n_families, planted)"""
    rng = np.random.default_rng(seed)
    if family_seed is None:
        fams = make_families(rng, n_tir, n_ltr)
    else:
        # population genomes (config C5): the families come from a seed all genomes share, each genome carries a random
        # family_keep fraction of them (its own copies, positions and divergences)
        fams = make_families(np.random.default_rng(family_seed), n_tir, n_ltr)
        for f_, keep_ in zip(fams, rng.random(len(fams)) < family_keep):
            if not keep_:
                f_["ncopy"] = 0
    n_chr = max(1, int(np.ceil(genome_bp / chrom_bp)))
    contig_off = np.minimum(np.arange(n_chr + 1, dtype=np.int64) * chrom_bp, genome_bp)
    # ---- planted copies ------------------------------------------------------------------------
    seqs, meta, divs = [], [], []  # meta: (family, strand, full_len_flag, tsd_len)
    for fi, fam in enumerate(fams):
        cons = fam["cons"]
        for k in range(fam["ncopy"]):
            div = float(rng.random() * 0.15) if k else 0.0
            s = _mutate_copy(rng, cons, div, 0.01 if k else 0.0)
            full = True
            if k and rng.random() < 0.2:  # 5' / 3' truncation
                cut = int(rng.integers(1, max(2, len(s) // 2)))
                full = cut < 0.05 * len(cons)
                s = s[cut:] if rng.random() < 0.5 else s[:-cut]
            minus = bool(rng.integers(0, 2))
            if minus:
                s = COMP[s[::-1]]
            seqs.append(s)
            meta.append((fi, minus, full, fam["tsd"]))
            divs.append(div)
    n_copies = len(seqs)
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    pad = 2 * flank + 40
    need = int((lens + pad).sum())
    if need > 0.8 * genome_bp:
        raise ValueError("planted copies (%d bp) do not fit the genome (%d bp)" % (need, genome_bp))
    order = rng.permutation(n_copies)
    free = genome_bp - need
    gaps = np.sort(rng.integers(0, free + 1, size=n_copies))
    starts = gaps + np.concatenate([[0], np.cumsum((lens + pad)[order])[:-1]]) + pad // 2
    pos = np.zeros(n_copies, dtype=np.int64)
    pos[order] = starts
    # drop copies that would cross a chromosome border (+- flank)
    chrom = np.searchsorted(contig_off, pos, side="right") - 1
    ok = (pos - pad // 2 >= contig_off[chrom]) & (pos + lens + pad // 2 <= contig_off[chrom + 1])
    # ---- background genome ---------------------------------------------------------------------
    if device is not None:
        import torch

        gen = torch.Generator(device=device)
        gen.manual_seed(int(seed))
        g = torch.empty(genome_bp + 64, dtype=torch.uint8, device=device)
        lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=device)
        thr = torch.tensor([0.29, 0.50, 0.71], device=device)
        step = 1 << 26
        for a in range(0, genome_bp, step):
            b = min(genome_bp, a + step)
            u = torch.rand(b - a, device=device, generator=gen)
            g[a:b] = lut[torch.bucketize(u, thr)]
        g[genome_bp:] = 65
        # overwrite with the planted copies (+ TSDs)
        buf = np.concatenate([ASCII[s] for s, o in zip(seqs, ok) if o]) if ok.any() else np.zeros(0, np.uint8)
        dst = np.concatenate([np.arange(p, p + l, dtype=np.int64) for p, l, o in zip(pos, lens, ok) if o]) if ok.any() else np.zeros(0, np.int64)
        tb = torch.from_numpy(buf).to(device)
        td = torch.from_numpy(dst).to(device)
        g[td] = tb
        del tb, td
        tsd_src, tsd_dst = [], []
        for p, l, o, m in zip(pos, lens, ok, meta):
            if o:
                t = m[3]
                tsd_src.append(np.arange(p - t, p, dtype=np.int64))
                tsd_dst.append(np.arange(p + l, p + l + t, dtype=np.int64))
        if tsd_src:
            ts = torch.from_numpy(np.concatenate(tsd_src)).to(device)
            tdd = torch.from_numpy(np.concatenate(tsd_dst)).to(device)
            g[tdd] = g[ts]
        genome = g

        def slice_bytes(a, b):
            return genome[a:b].cpu().numpy()
    else:
        gnp = ASCII[_rand_codes(rng, genome_bp)]
        gnp = np.concatenate([gnp, np.full(64, 65, np.uint8)])
        for s, p, l, o, m in zip(seqs, pos, lens, ok, meta):
            if o:
                gnp[p:p + l] = ASCII[s]
                t = m[3]
                gnp[p + l:p + l + t] = gnp[p - t:p]
        genome = gnp

        def slice_bytes(a, b):
            return genome[a:b]
    # ---- candidates + copy table ---------------------------------------------------------------
    crng = np.random.default_rng(seed + 7919 if cand_seed is None else cand_seed)
    by_fam = {}
    for i, (fi, minus, full, _t) in enumerate(meta):
        if ok[i] and full:
            by_fam.setdefault(fi, []).append(i)
    cand_chunks, cand_len, copy_first = [], [], [0]
    c_contig, c_s1, c_e1, c_minus, c_fam, c_div = [], [], [], [], [], []
    rc_lut = np.zeros(256, dtype=np.uint8)
    rc_lut[:] = ord("N")
    for a, b in zip(b"ACGT", b"TGCA"):
        rc_lut[a] = b
    for fi in sorted(by_fam):
        members = by_fam[fi]
        for v in range(cands_per_family):
            i = members[int(crng.integers(0, len(members)))]
            dl, dr = int(crng.integers(-30, 31)), int(crng.integers(-30, 31))
            a = int(pos[i]) - dl if not meta[i][1] else int(pos[i]) - dr
            b = int(pos[i] + lens[i]) + dr if not meta[i][1] else int(pos[i] + lens[i]) + dl
            a = max(a, int(contig_off[chrom[i]]))
            b = min(b, int(contig_off[chrom[i] + 1]))
            if b - a < 60:
                a, b = int(pos[i]), int(pos[i] + lens[i])
            sq = np.asarray(slice_bytes(a, b))
            if meta[i][1]:
                sq = rc_lut[sq[::-1]]
            cand_chunks.append(np.ascontiguousarray(sq))
            cand_len.append(len(sq))
            for j in members:
                f1, f2 = int(crng.integers(0, 3)), int(crng.integers(0, 3))
                c_contig.append(int(chrom[j]))
                c_s1.append(int(pos[j] - contig_off[chrom[j]]) + 1 - f1)
                c_e1.append(int(pos[j] - contig_off[chrom[j]] + lens[j]) + f2)
                c_minus.append(1 if meta[j][1] else 0)
            copy_first.append(len(c_contig))
            c_fam.append(fi)
            c_div.append(divs[i])
    cand_off = np.zeros(len(cand_len) + 1, dtype=np.int64)
    np.cumsum(cand_len, out=cand_off[1:])
    cands = np.concatenate(cand_chunks) if cand_chunks else np.zeros(0, np.uint8)
    return dict(genome=genome, genome_bp=genome_bp, contig_off=contig_off, cands=cands, cand_off=cand_off,
                copy_first=np.array(copy_first, dtype=np.int32), contig=np.array(c_contig, dtype=np.int32),
                start1=np.array(c_s1, dtype=np.int64), end1=np.array(c_e1, dtype=np.int64),
                minus=np.array(c_minus, dtype=np.uint8), family=np.array(c_fam, dtype=np.int32),
                cand_div=np.array(c_div, dtype=np.float64),   # divergence of the copy each candidate was cut from
               
                n_families=len(fams), n_planted=int(ok.sum()), planted_bp=int(lens[ok].sum()),
                fam_is_ltr=np.array([f["ltr"] for f in fams], dtype=bool),
                fam_len=np.array([len(f["cons"]) for f in fams], dtype=np.int64),
                # ground truth of the planted copies (tests / recall measurements): family, contig, 0-based start inside the
                # contig, length, strand, "full length" flag (not truncated beyond 5 %), divergence from the family consensus
                planted=dict(family=np.array([m[0] for m in meta], dtype=np.int32)[ok], contig=chrom[ok].astype(np.int32),
                             start=(pos - contig_off[chrom])[ok], length=lens[ok],
                             minus=np.array([m[1] for m in meta], dtype=bool)[ok], full=np.array([m[2] for m in meta], dtype=bool)[ok],
                             div=np.array(divs, dtype=np.float64)[ok]))
