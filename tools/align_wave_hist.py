#!/usr/bin/env python
"""Per-wavefront durations of the alignment kernels: is the time of align_fwd_kernel<4> /
align_tb_kernel the TAIL of their longest wavefronts or stalls inside the wavefronts?  Needs a library built with
-DALIGN_CLOCKS (every wavefront records start / end on the 100 MHz constant clock, the SIMD it ran on and the columns of
its longest lane).  One C3 fine-stage step; the clocks of the LAST launch of each kernel (pass B, the full-length windows:
10.7 of the 13.7 ms of the forward kernel, 10.7 of 13.8 of the traceback).
usage (GPU box, after building with -DALIGN_CLOCKS): python tools/align_wave_hist.py > profiles/r04_align_wave_hist.txt"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def analyse(name, a, out):
    t0, t1, hw, cols = a
    ok = t1 > 0
    if ok.any():
        # the arrays keep the entries of earlier launches at block indices the last launch did not reach: the last launch is the
        # group of wavefronts that started within 20 ms of the latest start (a launch lasts ~10 ms, launches of one kernel are > 40 ms apart)
        ok &= t0.astype(np.int64) >= int(t0[ok].max()) - 2_000_000
    t0, t1, hw, cols = t0[ok].astype(np.int64), t1[ok].astype(np.int64), hw[ok], cols[ok].astype(np.int64)
    if len(t0) == 0:
        out.append("%s: no wavefront recorded" % name)
        return
    base = t0.min()
    s, e = (t0 - base) * 0.01, (t1 - base) * 0.01          # microseconds
    d = e - s
    span = e.max()
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    se = (hw >> 13) & 7
    xcc = (hw >> 16) & 15
    slot = ((xcc * 8 + se) * 16 + cu) * 4 + simd
    nslot = len(np.unique(slot))
    # waves in flight over time
    ev = np.concatenate([np.stack([s, np.ones_like(s)], 1), np.stack([e, -np.ones_like(e)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    act = np.cumsum(ev[:, 1])
    dt = np.diff(np.concatenate([ev[:, 0], [span]]))
    peak = act.max()
    mean_act = float((act * dt).sum() / span)
    out.append("%s: %d wavefronts on %d SIMDs, kernel span %.0f us; waves in flight: peak %d, time-average %.0f (%.2f of the peak)" %
               (name, len(d), nslot, span, peak, mean_act, mean_act / peak))
    for frac in (0.75, 0.5, 0.25, 0.1):
        out.append("    time with fewer than %2.0f %% of the peak in flight: %5.1f %% of the span" % (100 * frac, 100.0 * dt[act < frac * peak].sum() / span))
    q = np.percentile(d, [5, 25, 50, 75, 95, 100])
    out.append("    wave duration (us): p5 %.0f p25 %.0f median %.0f p75 %.0f p95 %.0f max %.0f;  columns of the longest lane: median %d, max %d" %
               (q[0], q[1], q[2], q[3], q[4], q[5], int(np.median(cols)), int(cols.max())))
    # ns per column as a function of how many waves shared the SIMD while the wave ran
    per_col = 1000.0 * d / np.maximum(cols, 1)
    order = np.argsort(slot, kind="stable")
    share = np.zeros(len(d))
    for sl in np.unique(slot):
        idx = np.flatnonzero(slot == sl)
        ss, ee = s[idx], e[idx]
        for k, i in enumerate(idx):
            ov = np.clip(np.minimum(ee, e[i]) - np.maximum(ss, s[i]), 0, None).sum()       # includes the wave itself
            share[i] = ov / max(d[i], 1e-9)
    out.append("    ns per column of a wavefront by the average number of wavefronts on its SIMD while it ran:")
    for lo, hi in ((0, 1.5), (1.5, 2.5), (2.5, 3.5), (3.5, 4.5), (4.5, 5.5), (5.5, 99)):
        m = (share >= lo) & (share < hi)
        if m.sum() >= 5:
            out.append("        %.1f - %.1f waves: %6d wavefronts, median %.0f ns per column (p25 %.0f, p75 %.0f)" %
                       (lo, min(hi, 9.9), int(m.sum()), np.median(per_col[m]), np.percentile(per_col[m], 25), np.percentile(per_col[m], 75)))
    late = s > 0.6 * span
    if late.sum() >= 5:
        out.append("    wavefronts that START in the last 40 %% of the span: %d, median %.0f ns per column; those that start in the first 40 %%: %d, median %.0f" %
                   (int(late.sum()), np.median(per_col[late]), int((s < 0.4 * span).sum()), np.median(per_col[s < 0.4 * span])))


def main():
    import torch

    import hite_amd
    from test_gpu_scale import run_fine

    lib = C.CDLL(os.path.join(ROOT, "hite_amd", "libhite_gpu.so"))
    if not hasattr(lib, "hite_debug_align_clocks"):
        sys.exit("libhite_gpu.so was not built with -DALIGN_CLOCKS")
    R = run_fine(1000, 2500, 2500, 20250927 + 3)
    torch.cuda.synchronize()
    n = 1 << 17
    out = ["# tools/align_wave_hist.py -- config C3, one fine-stage step, the LAST launch of each kernel (pass B, full-length windows); MI355X, library built with -DALIGN_CLOCKS",
           "# per wavefront: start / end on the 100 MHz constant clock (s_memrealtime), HW_ID / XCC_ID, columns of its longest lane"]
    for which, name in ((0, "align_fwd_kernel<4>"), (1, "align_tb_kernel")):
        buf = np.zeros(4 * n, dtype=np.uint64)
        rc = lib.hite_debug_align_clocks(which, buf.ctypes.data_as(C.c_void_p), n)
        assert rc == 0
        analyse(name, buf.reshape(4, n), out)
    R["ctx"].close()
    print("\n".join(out))


if __name__ == "__main__":
    main()
