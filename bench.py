#!/usr/bin/env python
"""bench.py -- candidate TE boundaries / second through the MI355X-native fine (dynamic-boundary)
stage on a synthetic genome (BASELINE.json metric; config C3 by default: 1 Gbp, ~50k mixed
LTR/TIR candidates).

One "step" = one pass of the hot path over the whole candidate batch:
  copy table -> window rules / row selection -> flank gather from the resident 2-bit genome ->
  star alignment -> sparse-column removal -> judge_boundary_v5 (first500+last500 pass first for
  >1 kb windows, then the full pass), inputs resident in HBM when the timed region starts.
Multi-GPU (weak scaling): the genome is replicated, every rank judges its own candidate batch and
the 32-byte call records are all-gathered over RCCL inside the timed step.

Prints ONE JSON line (rank 0).  See DESIGN.md section "Measurement" for the byte accounting.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
PEAK_VALU_GINST = 256 * 4 * 2.4 / 4      # G wave64 vector instructions per second (MI355X_MICROARCH.md: 4 cycles per wave64 VALU op)
VALU_PER_DP_STEP = 11.6                  # measured: SQ_INSTS_VALU / (launch steps), profiles/r01_sq_counters.txt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genome-mbp", type=int, default=int(os.environ.get("HITE_BENCH_MBP", 1000)))
    ap.add_argument("--tir-families", type=int, default=None)
    ap.add_argument("--ltr-families", type=int, default=None)
    ap.add_argument("--cands-per-family", type=int, default=10)
    ap.add_argument("--seed", type=int, default=20250927 + 3)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=1, help="worker processes of the cpu_baseline leg (opt-in; default: the scalar port on one core)")
    ap.add_argument("--copies", choices=["found", "truth"], default="found",
                    help="found: copy finding (minimizer index lookup) runs inside the timed step; truth: the generator's copy table is the input")
    ap.add_argument("--verify", type=int, default=0, help="re-judge this many random candidates with the CPU oracle chain and compare")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("HITE_BENCH_STREAMS", 1)),
                    help="library contexts (each with its own HIP stream and share of the candidates) driven concurrently on every GPU")
    ap.add_argument("--stage", choices=["fine", "coarse"], default="fine",
                    help="fine (default): BASELINE.json's metric; coarse: the companion line of stage 3.1 (all-vs-all seeding + FMEA)")
    args = ap.parse_args()
    if args.stage == "coarse":
        return coarse_stage(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import hite_amd
    from hite_amd import synth
    from hite_amd._lib import CALL_DTYPE

    G = args.genome_mbp * 1_000_000
    # C3 proportions: 2.5 families of each kind per Mbp (2.5k TIR + 2.5k LTR on 1 Gbp)
    n_tir = args.tir_families if args.tir_families is not None else max(1, int(2.5 * args.genome_mbp))
    n_ltr = args.ltr_families if args.ltr_families is not None else max(0, int(2.5 * args.genome_mbp))
    t0 = time.time()
    # same genome on every rank (replicated), rank-specific candidate draw (weak scaling)
    w = synth.make_workload(genome_bp=G, n_tir=n_tir, n_ltr=n_ltr, cands_per_family=args.cands_per_family, seed=args.seed,
                            device=dev, cand_seed=args.seed + 7919 + 104729 * rank)
    setup_s = time.time() - t0
    n_cand = len(w["cand_off"]) - 1
    n_copies = len(w["contig"])

    # K library contexts on this GPU, each with its own HIP stream, its own copy of the packed genome + index and a
    # contiguous share of the candidates, driven by K host threads: while one share is in the vector-ALU-bound star
    # alignment the other runs its memory / latency-bound stages (copy finding, fill, judges).
    K = max(1, args.streams)
    ctxs = [hite_amd.Context(local_rank) for _ in range(K)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
    sptr = [st_.cuda_stream for st_ in streams]
    index_s = 0.0
    for ctx_, sp_ in zip(ctxs, sptr):
        ctx_.genome_pack_dev(w["genome"].data_ptr(), w["contig_off"], sp_)
    torch.cuda.synchronize()
    if args.copies == "found":
        ti = time.time()
        for ctx_, sp_ in zip(ctxs, sptr):
            ctx_.copy_index_build(sp_)   # once per genome (like `minimap2 -d`, Util.py:7941): part of genome residency, untimed
        torch.cuda.synchronize()
        index_s = (time.time() - ti) / K
    ctx = ctxs[0]

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    d_calls = torch.zeros(n_cand * 32, dtype=torch.uint8, device=dev)
    cons_cap = int(w["cand_off"][-1]) + 200 * n_cand + 4096 * K
    d_cons = torch.zeros(cons_cap + 64 * K, dtype=torch.uint8, device=dev)
    gathered = torch.zeros(world * n_cand * 32, dtype=torch.uint8, device=dev) if world > 1 else None
    cand_bytes = int(w["cand_off"][-1])
    # uneven shares on purpose when K > 1: identical shares would march through their phases in lockstep and never overlap a
    # vector-bound phase of one with a memory-bound phase of the other
    wts = [1.0 + 0.35 * i for i in range(K)]
    acc = np.cumsum([0.0] + wts) / sum(wts)
    bounds = [int(round(n_cand * a_)) for a_ in acc]
    parts = []
    for i in range(K):
        lo, hi = bounds[i], bounds[i + 1]
        b0, b1 = int(w["cand_off"][lo]), int(w["cand_off"][hi])
        c0, c1 = int(w["copy_first"][lo]), int(w["copy_first"][hi])
        cons_base = b0 + 200 * lo + 4096 * i
        parts.append({
            "lo": lo, "n": hi - lo, "bytes": b1 - b0, "n_copies": c1 - c0,
            "cand": up(np.concatenate([w["cands"][b0:b1], np.zeros(64, np.uint8)])), "cand_off": up(w["cand_off"][lo:hi + 1] - b0),
            "cf": up(w["copy_first"][lo:hi + 1] - c0), "contig": up(w["contig"][c0:c1]), "s1": up(w["start1"][c0:c1]),
            "e1": up(w["end1"][c0:c1]), "mn": up(np.concatenate([w["minus"][c0:c1], np.zeros(16, np.uint8)])),
            "calls_ptr": d_calls.data_ptr() + 32 * lo, "cons_ptr": d_cons.data_ptr() + cons_base, "cons_base": cons_base,
            "cons_cap": (b1 - b0) + 200 * (hi - lo) + 4096, "found": None, "stats": None, "err": None})
    found = {"n": n_copies}

    def run_part(i):
        P, cx, sp_ = parts[i], ctxs[i], sptr[i]
        try:
            torch.cuda.set_device(local_rank)
            if P["n"] == 0:
                P["stats"] = np.zeros(12, dtype=np.int64)
                return
            if args.copies == "found":
                nc, p_cf, p_ct, p_s1, p_e1, p_mn, _p_an = cx.find_copies_dev(P["n"], P["cand"].data_ptr(), P["cand_off"].data_ptr(), P["bytes"], sp_)
                P["found"] = (nc, p_cf, p_ct, p_s1, p_e1, p_mn)
                P["stats"] = cx.flank_region_align_dev("tir", 1, P["n"], P["cand"].data_ptr(), P["cand_off"].data_ptr(), p_cf, nc, p_ct, p_s1,
                                                       p_e1, p_mn, 50, P["calls_ptr"], P["cons_ptr"], P["cons_cap"], sp_)
            else:
                P["stats"] = cx.flank_region_align_dev("tir", 1, P["n"], P["cand"].data_ptr(), P["cand_off"].data_ptr(), P["cf"].data_ptr(),
                                                       P["n_copies"], P["contig"].data_ptr(), P["s1"].data_ptr(), P["e1"].data_ptr(),
                                                       P["mn"].data_ptr(), 50, P["calls_ptr"], P["cons_ptr"], P["cons_cap"], sp_)
        except Exception as e:  # noqa: BLE001  (re-raised on the main thread)
            P["err"] = e

    import threading

    def step():
        if K == 1:
            run_part(0)
        else:
            th = [threading.Thread(target=run_part, args=(i,)) for i in range(K)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
        for P in parts:
            if P["err"] is not None:
                raise P["err"]
        for st_ in streams:
            st_.synchronize()
        if args.copies == "found":
            found["n"] = sum(P["found"][0] for P in parts if P["found"])
        if world > 1:
            dist.all_gather_into_tensor(gathered, d_calls)  # merge the boundary calls (RCCL over xGMI)
        return sum(P["stats"] for P in parts)

    # residency set-up, like the index build above: the library's grow-only arenas reach their final size in the first call
    # and are consolidated into one block at the start of the second (hite_arena.h); from the third call on a step performs
    # no hipMalloc / hipFree.  These two calls are not warm-up steps of the measurement (they run whatever --warmup says).
    for _ in range(2):
        step()
    for _ in range(args.warmup):
        step()
    for ctx_ in ctxs:
        ctx_.profile(on=True, reset=True)
        ctx_.align_stats(reset=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    stats = None
    if K == 1:
        for _ in range(args.steps):
            stats = step()
    else:
        # the shares run their K steps back to back on their own stream / host thread (no join between steps: a step of a
        # share is an independent unit of work); the boundary calls of every step are gathered afterwards
        def run_steps(i):
            for _ in range(args.steps):
                run_part(i)
                if parts[i]["err"] is not None:
                    return

        th = [threading.Thread(target=run_steps, args=(i,)) for i in range(K)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        for P in parts:
            if P["err"] is not None:
                raise P["err"]
        for st_ in streams:
            st_.synchronize()
        if args.copies == "found":
            found["n"] = sum(P["found"][0] for P in parts if P["found"])
        if world > 1:
            for _ in range(args.steps):
                dist.all_gather_into_tensor(gathered, d_calls)
        stats = sum(P["stats"] for P in parts)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t1
    prof = {}
    for ctx_ in ctxs:   # per-stage HIP-event times of every context (kernels of different contexts may overlap in time)
        for k_, (ms_, cnt_) in ctx_.profile(on=False).items():
            a_, b_ = prof.get(k_, (0.0, 0))
            prof[k_] = (a_ + ms_, b_ + cnt_)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    align_stats = {}
    for ctx_ in ctxs:
        for k_, v_ in ctx_.align_stats().items():
            align_stats[k_] = v_ if k_ == "exact_cap" else align_stats.get(k_, 0) + v_
    calls = d_calls.cpu().numpy().view(CALL_DTYPE).copy()
    for P in parts:   # consensus offsets are relative to each share's pool
        calls["cons_off"][P["lo"]:P["lo"] + P["n"]] += P["cons_base"]
    n_te = int((calls["is_te"] != 0).sum())

    if rank == 0:
        ms_per_step = 1000.0 * elapsed / max(1, args.steps)
        value = world * n_cand * args.steps / elapsed
        # ---- roofline of the dominant kernel (HIP events recorded by the library on its launch stream) ----
        rows = int(stats[0] + stats[4])
        alg = {
            # bytes per STEP (all launches of that kernel in one step); see DESIGN.md "Measurement"
            "star_align_kernel": float(stats[3] + stats[7]),
            "row_gather_kernel": float((stats[1] + stats[5]) * (1 + 0.375)),
            "star_layout_kernel": float(2 * 0.5 * (stats[3] + stats[7]) / 3.0),
            "star_fill_sparse_kernel": float((stats[1] + stats[5]) + (stats[2] + stats[6])),
            "judge_kernel": float((stats[2] + stats[6]) * (1 + 13.0 / 32.0)),
            "select_rows_kernel": float(8 * n_copies + 4 * rows),
        }
        kern = {}
        for k, (ms, cnt) in prof.items():
            kern[k] = {"ms_per_step": ms / args.steps, "launches_per_step": cnt / args.steps}
        # the library times the two passes of a step separately; fold them per kernel for the roofline
        merged = {}
        for k, (ms, cnt) in prof.items():
            base = k.replace("_passA", "").replace("_passB", "").replace("_long", "").replace("_short", "")
            a, b = merged.get(base, (0.0, 0))
            merged[base] = (a + ms, b + cnt)
        prof = merged
        dom = max(prof.items(), key=lambda kv: kv[1][0])[0] if prof else None
        roof = None
        if dom:
            ms_tot, cnt = prof[dom]
            avg_launch_ms = ms_tot / max(1, cnt)
            bytes_per_launch = alg.get(dom, 0.0) * args.steps / max(1, cnt)
            achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get(dom, {}).get("bytes_per_launch")
                except Exception:
                    traffic = None
            roof = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 3), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(achieved / PEAK_HBM_GBS, 6), "traffic": traffic,
                    "avg_launch_ms": round(avg_launch_ms, 4), "alg_bytes_per_launch": int(bytes_per_launch)}
            if dom == "star_align_kernel":
                cells = 64.0 * float(stats[8] + stats[9])
                # the kernel is bound by vector-instruction issue (profiles/r01_sq_counters.txt: SQ_INSTS_VALU per anti-diagonal
                # step of a wave = 11.6 incl. traceback, SIMD busy ~98 %): report that rate beside the HBM figure.
                # peak = 256 CU x 4 SIMD x 2.4 GHz / 4 cycles per wave64 instruction
                steps_done = float(stats[8] + stats[9]) * args.steps
                valu_rate = VALU_PER_DP_STEP * steps_done / (ms_tot * 1e-3) / 1e9
                roof["note"] = ("vector-issue-bound banded DP (not HBM): traffic is the traceback's 2 direction bits per cell, "
                                "written once, read once (16 B per wave-step)")
                roof["dp_gcells_per_s"] = round(cells * args.steps / (ms_tot * 1e-3) / 1e9, 2)
                roof["valu_issue"] = {"achieved": round(valu_rate, 1), "peak": PEAK_VALU_GINST, "unit": "G wave64-inst/s",
                                      "frac": round(valu_rate / PEAK_VALU_GINST, 4), "inst_per_dp_step": VALU_PER_DP_STEP}
        out = {
            "metric": "candidate TE boundaries/sec on %s synthetic genome (fine stage: copy finding+gather+align+vote+judge)" % ("1 Gbp" if args.genome_mbp == 1000 else "%d Mbp" % args.genome_mbp),
            "value": round(value, 2), "unit": "candidates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "C3: %d Mbp synthetic genome, %d TIR + %d LTR families, %d candidates/GPU judged as TIR "
                                   "(%s)" % (args.genome_mbp, n_tir, n_ltr, n_cand,
                                         "copy finding by minimizer-index lookup inside the timed step; index build %.1f s untimed" % index_s
                                         if args.copies == "found" else "copy table = generator truth; copy finding not in the timed path"),
                       "genome_bp": G, "candidates_per_gpu": n_cand, "candidate_bases": cand_bytes, "copies": int(found["n"]), "copy_table": args.copies, "rows_aligned_per_step": rows,
                       "pipeline_stats": [int(x) for x in stats], "copy_stats": [int(x) for x in np.sum([cx.copy_stats() for cx in ctxs], axis=0)] if args.copies == "found" else None,
                       "streams": K, "align_stats_per_step": {k_: (v_ // max(1, args.steps) if k_ != "exact_cap" else v_) for k_, v_ in align_stats.items()},
                       "is_te": n_te, "parallelism": "replicated genome, candidates sharded x%d, all-gather of 32-B calls" % world,
                       "setup_s": round(setup_s, 1)},
            "roofline": roof,
            "kernels": {k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in sorted(kern.items())},
        }
        if hasattr(ctx.lib, "hite_debug_judge_clocks"):   # only in a -DJUDGE_CLOCKS development build
            import ctypes
            buf = (ctypes.c_ulonglong * 16)()
            ctx.lib.hite_debug_judge_clocks(buf, 1)
            out["judge_phase_ticks"] = [int(x) for x in buf[:12]]
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(w, args.cpu_seconds, args.cpu_threads)
            except Exception as e:   # the GPU line must not depend on the CPU leg
                out["cpu_baseline"] = {"value": None, "unit": "candidates/s", "cores": args.cpu_threads, "kind": "port",
                                       "sample": "failed: %s: %s" % (type(e).__name__, e)}
        if args.verify > 0:
            if args.copies == "found":
                # the oracle chain re-judges on the SAME copy table the GPU found (copy finding itself is
                # checked against its twin in tests/test_gpu_parity.py::test_find_copies_vs_twin)
                w = dict(w)
                cf_all, ct_all, s1_all, e1_all, mn_all, base = [np.zeros(1, np.int32)], [], [], [], [], 0
                for P, cx in zip(parts, ctxs):
                    if not P["found"]:
                        continue
                    nc, p_cf, p_ct, p_s1, p_e1, p_mn = P["found"]
                    cf_all.append(cx.download(p_cf, P["n"] + 1, np.int32)[1:] + base)
                    ct_all.append(cx.download(p_ct, nc, np.int32)); s1_all.append(cx.download(p_s1, nc, np.int64))
                    e1_all.append(cx.download(p_e1, nc, np.int64)); mn_all.append(cx.download(p_mn, nc, np.uint8))
                    base += nc
                w["copy_first"] = np.concatenate(cf_all)
                w["contig"], w["start1"], w["end1"], w["minus"] = (np.concatenate(x) for x in (ct_all, s1_all, e1_all, mn_all))
            out["verify"] = verify(w, calls, d_cons.cpu().numpy(), args.verify)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def verify(w, calls, cons, count):
    """full-size parity spot check: the oracle chain on random candidates of THIS workload vs the GPU calls"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_pipeline as OP

    co = w["contig_off"]
    g = w["genome"]
    host = g.cpu().numpy() if hasattr(g, "cpu") else g
    contigs = {ci: host[co[ci]:co[ci + 1]].tobytes() for ci in range(len(co) - 1)}
    n_cand = len(w["cand_off"]) - 1
    rng = np.random.default_rng(12345)
    bad = []
    info_names = {0: "", 1: "nb", 2: "fl1", 3: "EXC"}
    picks = rng.permutation(n_cand)[:count]
    for c in picks:
        a, b = int(w["copy_first"][c]), int(w["copy_first"][c + 1])
        copies = [(int(w["contig"][i]), int(w["start1"][i]), int(w["end1"][i]), int(w["minus"][i])) for i in range(a, b)]
        cand = w["cands"][w["cand_off"][c]:w["cand_off"][c + 1]].tobytes().decode()
        exp = OP.fine_stage_candidate("tir", cand, copies, contigs, plant=1)
        r = calls[c]
        got = [bool(r["is_te"]), info_names[int(r["info"])],
               cons[r["cons_off"]:r["cons_off"] + r["cons_len"]].tobytes().decode() if r["is_te"] else "", int(r["row_num"])]
        if got != exp:
            bad.append(int(c))
    return {"checked": int(len(picks)), "mismatches": len(bad), "bad_candidates": bad[:10]}


def coarse_stage(args):
    """companion measurement of stage 3.1 (coarse_boundary.py: all-vs-all search of the 1 Mbp segments + FMEA) on the same
    synthetic genome; one JSON line in the same shape as the main line.  A step = index + hite_seed_allvsall + hite_fmea_chain
    over the whole genome (one chunk).  cpu_baseline = the CPU twins of the same stages (orc_seed_allvsall + orc_fmea, single
    thread) on a bounded sub-genome (the all-vs-all stage is super-linear in the genome, so Mbp/s on the sample flatters
    the CPU)."""
    import torch

    import hite_amd
    from hite_amd import synth

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    G = args.genome_mbp * 1_000_000
    n_tir = args.tir_families if args.tir_families is not None else max(1, int(2.5 * args.genome_mbp))
    n_ltr = args.ltr_families if args.ltr_families is not None else max(0, int(2.5 * args.genome_mbp))
    # replicas: every rank searches its own genome (config 5 of BASELINE.json: one genome per GPU); no collective on the data path
    w = synth.make_workload(genome_bp=G, n_tir=n_tir, n_ltr=n_ltr, cands_per_family=1, seed=args.seed + 977 * rank, device=dev)
    ctx = hite_amd.Context(local_rank)
    ctx.genome_pack_dev(w["genome"].data_ptr(), w["contig_off"])
    sc, so = ctx.seed_segments(1_000_000)

    def step():
        ctx.copy_index_build()                      # the index is part of the step here (rebuilt on the same handle)
        (oc, _os, _oe), st = ctx.coarse_stage_dev(1_000_000, sc, so, 4000, 30000)   # the HSP table never leaves the device
        return st, len(oc)

    # two untimed steps at least: the first grows the arenas of the index state, the second consolidates them into one block
    args.warmup = max(args.warmup, 2)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        stats, n_iv = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t1
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        ms = 1000.0 * elapsed / max(1, args.steps)
        out = {"metric": "coarse_boundary step (stage 3.1: all-vs-all seeding + FMEA) on the 1 Gbp synthetic genome",
               "value": round(world * args.genome_mbp * args.steps / elapsed, 2), "unit": "Mbp/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "u8", "data": "synthetic",
               "config": {"workload": "C3 genome: %d Mbp, %d TIR + %d LTR families, 1 Mbp segments, one chunk%s" %
                                      (args.genome_mbp, n_tir, n_ltr, "; one genome per GPU (replicas)" if world > 1 else ""),
                          "seeds": stats[0], "anchors": stats[1], "clusters": stats[2], "hsp_records": stats[3], "repeat_intervals": n_iv},
               "roofline": {"bound": "hbm", "achieved": round((12.0 * stats[0] * 9 + 24.0 * stats[1] * 5 + 48.0 * stats[3] * 5) / (elapsed / args.steps) / 1e9, 2),
                            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                            "note": "algorithmic bytes = radix passes x 2 x record size over seeds / anchors / HSP records (whole step, not one kernel)"}}
        out["roofline"]["frac"] = round(out["roofline"]["achieved"] / PEAK_HBM_GBS, 5)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = coarse_cpu_baseline(args, min(args.genome_mbp, 20))
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def coarse_cpu_baseline(args, mbp):
    """the CPU twins of the coarse stage (oracle/hite_oracle_copies.c: orc_seed_allvsall, oracle/hite_oracle_coarse.c: orc_fmea),
    single thread, on a sub-genome of `mbp` Mbp generated with the same family density"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from hite_amd import synth

    w = synth.make_workload(genome_bp=mbp * 1_000_000, n_tir=max(1, int(2.5 * mbp)), n_ltr=int(2.5 * mbp), cands_per_family=1, seed=args.seed)
    co = w["contig_off"]
    contigs = [w["genome"][co[i]:co[i + 1]].tobytes() for i in range(len(co) - 1)]
    t0 = time.perf_counter()
    h = O.seed_allvsall(contigs, seg_len=1_000_000)
    h["chrom_names"] = ["c%d" % i for i in range(len(contigs))]
    names = O.fmea(h, 4000, 30000)
    dt = time.perf_counter() - t0
    return {"value": round(mbp / dt, 3), "unit": "Mbp/s", "cores": 1, "kind": "port",
            "sample": "%d Mbp sub-genome, same family density (%.1f s): %d HSP records -> %d intervals; CPU twins of the same stages, single thread"
                      % (mbp, dt, len(h["qseg"]), len(names))}


_CPU = {}


def _cpu_worker(job):
    """one worker of the multi-process CPU baseline: judges its slice of the sample until the budget runs out"""
    cands, budget_s = job
    import oracle_pipeline as OP

    w, contigs = _CPU["w"], _CPU["contigs"]
    t0 = time.perf_counter()
    done = 0
    for c in cands:
        a, b = int(w["copy_first"][c]), int(w["copy_first"][c + 1])
        copies = [(int(w["contig"][i]), int(w["start1"][i]), int(w["end1"][i]), int(w["minus"][i])) for i in range(a, b)]
        cand = w["cands"][w["cand_off"][c]:w["cand_off"][c + 1]].tobytes().decode()
        OP.fine_stage_candidate("tir", cand, copies, contigs, plant=1)
        done += 1
        if time.perf_counter() - t0 > budget_s and done >= 8:
            break
    return done


def cpu_baseline(w, budget_s, threads=1):
    """the oracle chain (oracle/*.c through tests/oracle_pipeline.py: a CPU port of the same step, gather + alignment +
    sparse columns + judge on the copy table of the workload) timed on a bounded sample of the same candidates on this host.
    threads > 1 (--cpu-threads, opt-in): the sample is split over forked worker processes that only run CPU code -- how the
    reference itself fans candidates out (ProcessPoolExecutor, Util.py:8141)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_pipeline as OP  # noqa: F401  (imported before forking)

    co = w["contig_off"]
    g = w["genome"]
    host = g.cpu().numpy() if hasattr(g, "cpu") else g
    contigs = {ci: host[co[ci]:co[ci + 1]].tobytes() for ci in range(len(co) - 1)}
    hw = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in w.items() if k in ("copy_first", "contig", "start1", "end1", "minus", "cands", "cand_off")}
    _CPU["w"], _CPU["contigs"] = hw, contigs
    n_cand = len(hw["cand_off"]) - 1
    order = np.random.default_rng(1).permutation(n_cand)
    threads = max(1, int(threads))
    t0 = time.perf_counter()
    if threads == 1:
        done = _cpu_worker((order, budget_s))
    else:
        import multiprocessing as mp

        with mp.get_context("fork").Pool(threads) as pool:
            done = sum(pool.map(_cpu_worker, [(order[k::threads], budget_s) for k in range(threads)]))
    dt = time.perf_counter() - t0
    return {"value": round(done / dt, 3), "unit": "candidates/s", "cores": threads, "kind": "port",
            "sample": "%d random candidates of the same workload (%.1f s), oracle chain, %s" %
                      (done, dt, "single thread" if threads == 1 else "%d worker processes" % threads)}


if __name__ == "__main__":
    main()
