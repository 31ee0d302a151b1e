#!/usr/bin/env python
"""Throughput of the terminal-inverted-repeat stage (hite_itr_search, where the reference runs tools/itrsearch -i 0.7 -l 7) on the
GPU, with the CPU twin on one core beside it and -- in the build container, where /root/reference exists -- the tool itself.
    python tools/itr_bench.py [n_records]
Prints one JSON line: records/s of the device call (records already in HBM, HIP events around the launch), of the host call
(upload + kernel + download), of the twin."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch

    import casegen
    import hite_amd
    import oracle_lib as O

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    base = casegen.make_itr_cases(7001, 20000)
    seqs = [base[i % len(base)] for i in range(n)]
    ctx = hite_amd.Context(0)
    t0 = time.perf_counter()
    out = ctx.itr_search(seqs, end_len=40)
    host_s = time.perf_counter() - t0
    # device-resident: the same records in one buffer, HIP events around the launch
    sb = [s.encode() for s in seqs]
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(s) for s in sb], out=off[1:])
    d_buf = torch.from_numpy(np.frombuffer(b"".join(sb) + b"\0" * 16, dtype=np.uint8).copy()).cuda()
    d_off = torch.from_numpy(off).cuda()
    d_out = torch.zeros(n * 8, dtype=torch.int32, device="cuda")
    lib = ctx.lib
    args = (ctx.h, C.c_int64(n), C.c_void_p(d_buf.data_ptr()), C.c_void_p(d_off.data_ptr()), 40, 40, C.c_double(0.7), 7, 10, 16, 32, 32,
            C.c_void_p(d_out.data_ptr()), C.c_void_p(0))
    assert lib.hite_itr_search_dev(*args) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # (the library launches on the stream it is given: the null stream here, which torch's default stream is)
    e0.record()
    for _ in range(5):
        assert lib.hite_itr_search_dev(*args) == 0
    e1.record()
    torch.cuda.synchronize()
    dev_ms = e0.elapsed_time(e1) / 5
    assert np.array_equal(d_out.cpu().numpy().reshape(n, 8), out)
    m = min(n, 20000)
    t0 = time.perf_counter()
    tw = O.itr_search(seqs[:m], 40)
    twin_s = time.perf_counter() - t0
    assert np.array_equal(tw, out[:m])
    line = {"records": n, "found": int(out[:, 5].sum()), "device_ms": round(dev_ms, 3), "device_records_per_s": round(n / (dev_ms * 1e-3)),
            "host_call_records_per_s": round(n / host_s), "twin_one_core_records_per_s": round(m / twin_s),
            "dp_cells_per_s": round(n * 1600 / (dev_ms * 1e-3)), "note": "first 40 + last 40 bases per record: 40 x 40 cells, 103 skewed steps per wavefront"}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
