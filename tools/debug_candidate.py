#!/usr/bin/env python
"""Debug helper (GPU box): rebuild the bench workload, run ONE candidate stage by stage through the HIP
path and the oracle, report the first stage that differs and dump the inputs for local replay."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch

    import hite_amd
    import oracle_lib as O
    import oracle_pipeline as OP
    from hite_amd import synth

    mbp = int(sys.argv[1])
    cands = [int(x) for x in sys.argv[2].split(",")]
    seed = 20250927 + 3
    dev = torch.device("cuda", 0)
    w = synth.make_workload(genome_bp=mbp * 1_000_000, n_tir=int(2.5 * mbp), n_ltr=int(2.5 * mbp), cands_per_family=10, seed=seed,
                            device=dev, cand_seed=seed + 7919)
    ctx = hite_amd.Context(0)
    ctx.genome_pack_dev(w["genome"].data_ptr(), w["contig_off"], 0)
    torch.cuda.synchronize()
    host = w["genome"].cpu().numpy()
    co = w["contig_off"]
    contigs = {ci: host[co[ci]:co[ci + 1]].tobytes() for ci in range(len(co) - 1)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for c in cands:
        a, b = int(w["copy_first"][c]), int(w["copy_first"][c + 1])
        copies = [(int(w["contig"][i]), int(w["start1"][i]), int(w["end1"][i]), int(w["minus"][i])) for i in range(a, b)]
        cand = w["cands"][w["cand_off"][c]:w["cand_off"][c + 1]].tobytes().decode()
        exp = OP.fine_stage_candidate("tir", cand, copies, contigs, plant=1)
        got, stats = ctx.flank_region_align("tir", [cand], [copies], plant=1)
        print("cand", c, "len", len(cand), "copies", len(copies), "GPU", got[0][:2], got[0][3], len(got[0][2]), "ORACLE", exp[:2], exp[3], len(exp[2]))
        # stage by stage
        wins, tr = ctx.flank_gather([x[0] for x in copies], [x[1] for x in copies], [x[2] for x in copies], [x[3] for x in copies], 50)
        full = [x.decode() for x in wins if x is not None]
        trunc = [x.decode() for x in tr if x is not None]
        for name, ws in (("trunc", trunc), ("full", full)):
            if not ws:
                continue
            keep = OP.select_rows([len(x) for x in ws])
            ws = [ws[i] for i in keep]
            gm = ctx.star_msa([ws])[0]
            om = O.star_msa(ws)
            same = gm is not None and om is not None and gm.shape == om.shape and np.array_equal(gm, om)
            print("  ", name, "rows", len(ws), "msa same:", same, None if gm is None else gm.shape, None if om is None else om.shape)
            if om is None or gm is None:
                continue
            gc = ctx.sparse_cols([gm])[0]
            kc = O.sparse_cols(om).astype(bool)
            oc = np.ascontiguousarray(om[:, kc])
            print("   sparse same:", gc.shape == oc.shape and np.array_equal(gc, oc))
            gj = ctx.judge("tir", [oc], [cand], plant=1)[0]
            oj, _ = O.judge("tir", oc, cand, 1)
            print("   judge same:", [gj[0], gj[1], gj[2], gj[3]] == oj, gj[:2], gj[3], oj[:2], oj[3])
            np.savez_compressed(os.path.join(ROOT, "gpurun_out", "dbg_%d_%s.npz" % (c, name)), wins=np.array(ws), cand=cand,
                                gpu_msa=gm, gpu_clean=gc)


if __name__ == "__main__":
    main()
