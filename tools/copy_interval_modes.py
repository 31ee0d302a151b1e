#!/usr/bin/env python
"""Config C2 under the two interval modes of the copy finder (hite_copy_config): the whole-candidate interval this build hands
on by default, and the aligned interval get_copies_minimap2 reports (Util.py:8026).  TE calls and how many of them have both
ends exactly on the planted element (tests/test_gpu_scale.py boundary_stats), + the cost of the step.
usage (GPU box): python tools/copy_interval_modes.py >> profiles/r04_scale_tests.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch

    from test_gpu_scale import boundary_stats, run_fine

    print("# tools/copy_interval_modes.py -- config C2 (100 Mbp, 500 TIR families, 5000 candidates), the fine stage under both interval modes of the copy finder")
    for mode, name in ((0, "whole-candidate intervals (HITE_COPY_INTERVAL=whole; the default of rounds 2-4)"), (1, "aligned intervals (Util.py:8026), rows padded by the clipped bases (the default since round 5)"),
                       (2, "aligned intervals WITHOUT clip words (the reference's own tuples): clips estimated by the probe (round 6; rounds 3-5: bare windows)")):
        os.environ["HITE_COPY_INTERVAL"] = "aligned" if mode else "whole"
        os.environ["HITE_TEST_NO_CLIP"] = "1" if mode == 2 else "0"     # (tests/test_gpu_scale.py run_fine: hand no clip words on)
        from hite_amd import _lib as hl

        hl.load().hite_copy_config(-1)             # take the mode from the environment again
        t0 = time.time()
        R = run_fine(100, 500, 0, 20250927 + 2)
        torch.cuda.synchronize()
        n_tir, called, checked, exact, near = boundary_stats(R)
        st = R["align"]
        print("%-75s copies %6d; TE calls %4d of %d; of %d checked: both ends exact %4d (%.3f), within 3 bp %4d (%.3f); wide fall-backs %d; %.1f s" %
              (name, len(R["found"]["contig"]), called, n_tir, checked, exact, exact / max(1, checked), near, near / max(1, checked),
               st["fallback"], time.time() - t0))
        R["ctx"].close()
    os.environ.pop("HITE_COPY_INTERVAL", None)
    os.environ.pop("HITE_TEST_NO_CLIP", None)


if __name__ == "__main__":
    main()
