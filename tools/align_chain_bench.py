"""Per-kernel time of the pairwise aligner on N pairs of ~L columns, thread-per-pair kernels against the lane-parallel ones
(hite_align_lanes): what ONE pair's dependent chain costs per column in each kernel (N = 1), and what a wavefront / a machine
full of them costs.  python tools/align_chain_bench.py [L] [N ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hite_amd  # noqa: E402
from test_align_oracle import make_pair  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 11000
NS = [int(x) for x in sys.argv[2:]] or [1, 16, 64, 1024, 16384]
ctx = hite_amd.Context(0)
rng = np.random.default_rng(5)
base = [make_pair(rng, L) for _ in range(8)]
for N in NS:
    groups = [[bytes(base[i % 8][0]), bytes(base[i % 8][1])] for i in range(N)]
    for lanes in (-2, 0):
        ctx.align_lanes(lanes)
        ctx.star_msa(groups)
        ctx.profile(on=True, reset=True)
        for _ in range(3):
            ctx.star_msa(groups)
        prof = ctx.profile()
        ctx.profile(on=False)
        al = {k: round(v[0] / max(v[1], 1), 3) for k, v in prof.items() if k.startswith("align")}
        cols = np.mean([len(b) for _, b in base])
        per = {k: round(v * 1e3 / cols, 4) for k, v in al.items() if "prep" not in k}
        print("L=%d N=%d lanes=%d  ms per launch %s  | us per column of one pair %s" % (L, N, lanes, al, per), flush=True)
ctx.align_lanes(-1)
