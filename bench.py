#!/usr/bin/env python
"""bench.py -- candidate TE boundaries / second through the MI355X-native fine (dynamic-boundary)
stage on a synthetic genome (BASELINE.json metric; config C3 by default: 1 Gbp, ~50k mixed
LTR/TIR candidates; --config C2: 100 Mbp, ~5k TIR candidates).

One "step" = one pass of the hot path over the whole candidate batch:
  copy finding (minimizer index lookup) -> window rules / row selection -> flank gather from the resident
  2-bit genome -> star alignment -> sparse-column removal -> judge_boundary_v5 (first500+last500 pass first
  for >1 kb windows, then the full pass), inputs resident in HBM when the timed region starts.
Multi-GPU: `python bench.py --gpus N` starts N ranks itself (one process per GPU, RCCL); under
torch.distributed.run it uses the ranks it is given.  The genome is replicated;
  --scaling weak  (default): every rank judges its own candidate batch (N x the work),
  --scaling strong: ONE batch is sharded over the ranks (config C4),
and the 32-byte call records are all-gathered over RCCL inside the timed step.

Prints ONE JSON line (rank 0).  See DESIGN.md section "Measurement" for the byte accounting.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
CONFIGS = {
    # name: (genome Mbp, TIR families per Mbp, LTR families per Mbp)  -- BASELINE.json configs[1] / configs[2]
    "C2": (100, 5.0, 0.0),
    "C3": (1000, 2.5, 2.5),
}


def valu_peak():
    """wave64 issue rate of the instruction classes the alignment kernels are made of, MEASURED on MI355X by
    tools/valu_issue_bench.hip (profiles/r02_valu_issue.txt): bit-field / 3-operand / carry / DPP instructions issue
    every ~4.1 cycles per SIMD, plain 2-operand add / xor every ~2.1 -> G wave-instructions/s at 8 waves per SIMD"""
    slow, fast = [], []
    try:
        for line in open(os.path.join(ROOT, "profiles", "r02_valu_issue.txt")):
            f = line.split()
            if len(f) >= 4 and f[-3] == "8":
                (fast if f[0] in ("v_add_u32", "v_xor_b32") and len(f) == 4 else slow).append(float(f[-2]))
    except OSError:
        pass
    slow = [x for x in slow if x > 300.0]   # drop the VCC-serialised v_cndmask line
    return (sum(slow) / len(slow) if slow else 600.0), (sum(fast) / len(fast) if fast else 1165.0)


def kept_counters():
    """per-kernel SQ counters of the committed profile (profiles/r02_sq_counters.json: rocprofv3 --pmc SQ_INSTS_VALU ...)"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "r02_sq_counters.json")))
    except (OSError, ValueError):
        return {}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=os.environ.get("HITE_BENCH_CONFIG", "C3"))
    ap.add_argument("--genome-mbp", type=int, default=None, help="override the genome size of the config (same family density)")
    ap.add_argument("--tir-families", type=int, default=None)
    ap.add_argument("--ltr-families", type=int, default=None)
    ap.add_argument("--cands-per-family", type=int, default=10)
    ap.add_argument("--seed", type=int, default=20250927 + 3)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="worker processes of the cpu_baseline leg (0 = min(40, cores of this host))")
    ap.add_argument("--cpu-copies", action="store_true",
                    help="cpu_baseline: also time the CPU twin of the copy finder on the whole genome (default up to 1 Gbp, where its index takes about a minute and a half)")
    ap.add_argument("--copies", choices=["found", "truth"], default="found",
                    help="found: copy finding (minimizer index lookup) runs inside the timed step; truth: the generator's copy table is the input")
    ap.add_argument("--verify", type=int, default=24, help="candidates re-judged with the CPU oracle chain after the timed region (0 = none)")
    ap.add_argument("--stage", choices=["fine", "coarse"], default="fine",
                    help="fine (default): BASELINE.json's metric; coarse: the companion line of stage 3.1 (all-vs-all seeding + FMEA)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # one process per GPU: start the ranks ourselves (the driver may equally well launch us under torch.distributed.run)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))
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
    from hite_amd import dist as hd
    from hite_amd import synth
    from hite_amd._lib import CALL_DTYPE

    mbp, tir_d, ltr_d = CONFIGS[args.config]
    if args.genome_mbp is not None:
        mbp = args.genome_mbp
    G = mbp * 1_000_000
    n_tir = args.tir_families if args.tir_families is not None else max(1, int(tir_d * mbp))
    n_ltr = args.ltr_families if args.ltr_families is not None else max(0, int(ltr_d * mbp))
    strong = args.scaling == "strong" and world > 1
    t0 = time.time()
    # same genome on every rank (replicated); weak scaling: rank-specific candidate draw, strong scaling: one draw, sharded
    w = synth.make_workload(genome_bp=G, n_tir=n_tir, n_ltr=n_ltr, cands_per_family=args.cands_per_family, seed=args.seed,
                            device=dev, cand_seed=args.seed + 7919 + (0 if strong else 104729 * rank))
    setup_s = time.time() - t0
    n_all = len(w["cand_off"]) - 1
    if strong:
        c0, c1, (b0, b1), (k0, k1) = hd.shard_candidates(w["cand_off"], w["copy_first"], rank, world)
    else:
        c0, c1, b0, b1, k0, k1 = 0, n_all, 0, int(w["cand_off"][-1]), 0, len(w["contig"])
    n_cand = c1 - c0

    ctx = hite_amd.Context(local_rank)
    stream = torch.cuda.Stream(device=dev)
    sp = stream.cuda_stream
    ctx.genome_pack_dev(w["genome"].data_ptr(), w["contig_off"], sp)
    torch.cuda.synchronize()
    index_s = 0.0
    if args.copies == "found":
        ti = time.time()
        ctx.copy_index_build(sp)   # once per genome (like `minimap2 -d`, Util.py:7941): part of genome residency, untimed
        torch.cuda.synchronize()
        index_s = time.time() - ti

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    d_calls = torch.zeros(max(1, n_cand) * 32, dtype=torch.uint8, device=dev)
    cons_cap = (b1 - b0) + 200 * n_cand + 4096
    d_cons = torch.zeros(cons_cap + 64, dtype=torch.uint8, device=dev)
    d_cand = up(np.concatenate([w["cands"][b0:b1], np.zeros(64, np.uint8)]))
    d_cand_off = up(w["cand_off"][c0:c1 + 1] - b0)
    d_cf = up(w["copy_first"][c0:c1 + 1] - k0)
    d_ct, d_s1, d_e1 = up(w["contig"][k0:k1]), up(w["start1"][k0:k1]), up(w["end1"][k0:k1])
    d_mn = up(np.concatenate([w["minus"][k0:k1], np.zeros(16, np.uint8)]))
    state = {"found": None, "n_copies": k1 - k0}
    n_merge = n_all if strong else world * n_cand

    def step():
        if n_cand > 0:
            if args.copies == "found":
                nc, p_cf, p_ct, p_s1, p_e1, p_mn, _p_an = ctx.find_copies_dev(n_cand, d_cand.data_ptr(), d_cand_off.data_ptr(), b1 - b0, sp)
                state["found"] = (nc, p_cf, p_ct, p_s1, p_e1, p_mn)
                state["n_copies"] = nc
                st = ctx.flank_region_align_dev("tir", 1, n_cand, d_cand.data_ptr(), d_cand_off.data_ptr(), p_cf, nc, p_ct, p_s1, p_e1, p_mn,
                                                50, d_calls.data_ptr(), d_cons.data_ptr(), cons_cap, sp)
            else:
                st = ctx.flank_region_align_dev("tir", 1, n_cand, d_cand.data_ptr(), d_cand_off.data_ptr(), d_cf.data_ptr(), k1 - k0,
                                                d_ct.data_ptr(), d_s1.data_ptr(), d_e1.data_ptr(), d_mn.data_ptr(), 50, d_calls.data_ptr(),
                                                d_cons.data_ptr(), cons_cap, sp)
        else:
            st = np.zeros(12, dtype=np.int64)
        stream.synchronize()
        merged = None
        if world > 1:   # merge the boundary calls (RCCL over xGMI): ONE all-gather of the 32-byte records
            merged = hd.allgather_calls(d_calls[: n_cand * 32], n_merge)
        return st, merged

    # residency set-up, like the index build above: the library's grow-only arenas reach their final size in the first call
    # and are consolidated into one block at the start of the second (hite_arena.h); from the third call on a step performs
    # no hipMalloc / hipFree.  These two calls are not warm-up steps of the measurement (they run whatever --warmup says).
    for _ in range(2):
        step()
    for _ in range(args.warmup):
        step()
    ctx.profile(on=True, reset=True)
    ctx.align_stats(reset=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    stats, merged = None, None
    for _ in range(args.steps):
        stats, merged = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t1
    prof = ctx.profile(on=False)
    align_stats = ctx.align_stats()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    calls = d_calls.cpu().numpy().view(CALL_DTYPE)[:n_cand].copy()
    n_te = int((calls["is_te"] != 0).sum())
    if world > 1:
        tn = torch.tensor([n_te], dtype=torch.int64, device=dev)
        dist.all_reduce(tn)
        n_te_all = int(tn.item())
        merged_te = int((merged.cpu().numpy().view(CALL_DTYPE)["is_te"] != 0).sum())
        assert merged_te == n_te_all, "all-gathered records disagree with the per-rank counts"
    else:
        n_te_all = n_te

    if rank == 0:
        steps = max(1, args.steps)
        ms_per_step = 1000.0 * elapsed / steps
        total_cands = n_all if strong else world * n_cand
        value = total_cands * args.steps / elapsed
        per_step = {k_: (v_ / steps if k_ != "exact_cap" else v_) for k_, v_ in align_stats.items()}
        rows = int(stats[0] + stats[4])
        cols_step = per_step["columns"]
        pairs_step = max(1.0, per_step["pairs"])
        kern = {k: {"ms_per_step": ms / steps, "launches_per_step": cnt / steps} for k, (ms, cnt) in prof.items()}
        # the library times the two passes of a step separately; fold them per kernel for the roofline
        merged_prof = {}
        for k, (ms, cnt) in prof.items():
            base = k.replace("_passA", "").replace("_passB", "").replace("_long", "").replace("_short", "")
            a, b = merged_prof.get(base, (0.0, 0))
            merged_prof[base] = (a + ms, b + cnt)
        # algorithmic bytes per STEP of each kernel (DESIGN.md "Measurement")
        ops_bytes = max(0.0, float(stats[3] + stats[7]) - 2.0 * cols_step)   # stats[3|7]: sum over pairs of (m + n + 2 (m + 1)); ~ 2 (m + 1) per pair stays
        alg = {
            "align_fwd4": 5.0 * cols_step,                 # row base read + 64-B check point per 16 columns
            "align_fwd_wide": 7.0 * cols_step * (per_step["wide"] / pairs_step),   # + 2 B of boundary information per column
            "align_tb": 4.0 * cols_step + ops_bytes,                              # the records back (they carry the row bases) + 2 B of ops per centre position
            "row_gather_kernel": float((stats[1] + stats[5]) * (1 + 0.375)),
            "star_layout_sparse_kernel": ops_bytes,
            "star_fill_sparse_kernel": float((stats[1] + stats[5]) + (stats[2] + stats[6])),
            "judge_kernel": float((stats[2] + stats[6]) * (1 + 13.0 / 32.0)),
        }
        if args.copies == "found":
            cs = [int(x) for x in ctx.copy_stats()]          # candidate minimizers, hits, clusters, accepted copies of the last step
            dbits = max(1, int(G + 65536).bit_length())
            cbits = max(1, int(max(1, n_cand) - 1).bit_length())
            passes = -(-dbits // 10) + -(-(1 + cbits) // 10)
            alg["radix_sort_hits"] = float(cs[1]) * (8 + 16) * passes     # 8-byte keys: histogram read + scatter read/write per pass
            alg["hit_kernel"] = float(cs[1]) * (8 + 8)                       # index entry gathered + packed hit written
            alg["cluster_flag_kernel"] = float(cs[1]) * (8 + 4)
            alg["cluster_acc_kernel"] = float(cs[1]) * (8 + 4 + 8)
        dom = max(merged_prof.items(), key=lambda kv: kv[1][0])[0] if merged_prof else None
        roof = None
        if dom:
            ms_tot, cnt = merged_prof[dom]
            avg_launch_ms = ms_tot / max(1, cnt)
            bytes_per_launch = alg.get(dom, 0.0) * steps / max(1, cnt)
            achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
            # HBM traffic of that kernel from the committed PMC passes (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 runs:
            # profiles/r02_pmc_traffic.json); a profiler stage can cover several kernel instantiations
            traffic = None
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")))
                names = {"align_fwd4": ["align_fwd_kernel<4>"], "align_tb": ["align_tb_kernel"],
                         "align_fwd_wide": ["align_fwd_kernel<8>", "align_fwd_kernel<16>", "align_fwd_kernel<32>"],
                         "radix_sort_hits": ["rs_scatter_staged_kernel", "rs_hist_kernel<10, 32>"]}.get(dom, [dom])
                tot = sum(pmc[k_]["bytes_per_launch"] * pmc[k_]["launches"] for k_ in names if k_ in pmc)
                runs = pmc.get(names[0], {}).get("launches", 0)
                traffic = int(tot / runs) if tot and runs else None      # per launch of the stage
            except Exception:
                traffic = None
            roof = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 3), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(achieved / PEAK_HBM_GBS, 6), "traffic": traffic,
                    "avg_launch_ms": round(avg_launch_ms, 4), "alg_bytes_per_launch": int(bytes_per_launch)}
            # the alignment kernels are bound by vector-instruction issue, not by HBM: report the whole group against the MEASURED
            # issue rate (tools/valu_issue_bench.hip -> profiles/r02_valu_issue.txt); instructions per column from the kept counters
            align_ms = sum(ms for k, (ms, _c) in merged_prof.items() if k.startswith("align_")) / steps
            if align_ms > 0 and cols_step > 0:
                slow, fast = valu_peak()
                cnt_file = kept_counters()
                inst = sum(v.get("valu_inst_per_step", 0.0) for k, v in cnt_file.items() if k.startswith("align_"))
                cols_file = cnt_file.get("_columns_per_step", 0.0)
                blk = {"ms_per_step": round(align_ms, 3), "pairs_per_step": int(per_step["pairs"]), "columns_per_step": int(cols_step),
                       "band_gcells_per_s": round(128.0 * cols_step / (align_ms * 1e-3) / 1e9, 1),
                       "certified_frac": round(per_step["certified"] / pairs_step, 4), "wide_frac": round(per_step["wide"] / pairs_step, 4),
                       "fallback_per_step": per_step["fallback"], "dropped_per_step": per_step["dropped"], "exact_cap": per_step["exact_cap"],
                       "peak_ginst_3op": round(slow, 1), "peak_ginst_2op": round(fast, 1)}
                if inst > 0 and cols_file > 0:
                    per_col = inst / cols_file            # wave-level instructions per pair-column
                    rate = per_col * cols_step / (align_ms * 1e-3) / 1e9
                    blk["valu_issue"] = {"achieved": round(rate, 1), "peak": round(slow, 1), "unit": "G wave64-inst/s", "frac": round(rate / slow, 4),
                                         "wave_inst_per_pair_column": round(per_col, 4), "source": "profiles/r02_sq_counters.json"}
                roof["align"] = blk
            # every stage with an algorithmic byte count, against the same HBM peak (what each is really bound by: DESIGN.md section 4)
            roof["stages"] = {k: {"ms_per_step": round(ms / steps, 3), "achieved": round(alg[k] / (ms / steps * 1e-3) / 1e9, 1),
                                  "frac": round(alg[k] / (ms / steps * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
                              for k, (ms, _c) in sorted(merged_prof.items()) if k in alg and ms > 0}
        out = {
            "metric": "candidate TE boundaries/sec on %s synthetic genome (fine stage: copy finding+gather+align+vote+judge)" % ("1 Gbp" if mbp == 1000 else "%d Mbp" % mbp),
            "value": round(value, 2), "unit": "candidates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %d Mbp synthetic genome, %d TIR + %d LTR families, %d candidates%s judged as TIR (%s)" %
                                   (args.config if args.genome_mbp is None else "custom", mbp, n_tir, n_ltr, total_cands if strong else n_cand,
                                    " sharded over %d GPUs" % world if strong else "/GPU",
                                    "copy finding by minimizer-index lookup inside the timed step; index build %.1f s untimed" % index_s
                                    if args.copies == "found" else "copy table = generator truth; copy finding not in the timed path"),
                       "genome_bp": G, "candidates_per_gpu": n_cand, "candidate_bases": b1 - b0, "copies": int(state["n_copies"]),
                       "copy_table": args.copies, "rows_aligned_per_step": rows,
                       "pipeline_stats": [int(x) for x in stats], "copy_stats": [int(x) for x in ctx.copy_stats_ext()] if args.copies == "found" else None,
                       "align_stats_per_step": {k_: int(v_) for k_, v_ in per_step.items()},
                       "is_te": n_te_all, "parallelism": "replicated genome, candidates %s x%d, all-gather of 32-B calls" %
                                                        ("sharded" if strong else "per rank", world),
                       "setup_s": round(setup_s, 1)},
            "roofline": roof,
            "kernels": {k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in sorted(kern.items())},
        }
        if hasattr(ctx.lib, "hite_debug_judge_clocks"):   # only in a -DJUDGE_CLOCKS development build
            import ctypes
            buf = (ctypes.c_ulonglong * 16)()
            ctx.lib.hite_debug_judge_clocks(buf, 1)
            out["judge_phase_ticks"] = [int(x) for x in buf[:12]]
        wv = None
        if (world == 1 and not args.no_cpu_baseline) or args.verify > 0:
            wv = host_workload(w, ctx, state if args.copies == "found" else None, c0, c1)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(wv, args.cpu_seconds, args.cpu_threads, args.cpu_copies or mbp <= 1000)
            except Exception as e:   # the GPU line must not depend on the CPU leg
                out["cpu_baseline"] = {"value": None, "unit": "candidates/s", "cores": args.cpu_threads, "kind": "port",
                                       "sample": "failed: %s: %s" % (type(e).__name__, e)}
        if args.verify > 0:
            # the oracle chain re-judges on the SAME copy table the GPU used (copy finding itself is checked against its twin
            # in tests/test_gpu_parity.py::test_find_copies_vs_twin); outside the timed region
            out["verify"] = verify(wv, calls, d_cons.cpu().numpy(), args.verify)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def host_workload(w, ctx, state, c0, c1):
    """host copy of this rank's share of the workload, with the copy table the GPU step used"""
    g = w["genome"]
    host = g.cpu().numpy() if hasattr(g, "cpu") else np.asarray(g)
    b0 = int(w["cand_off"][c0])
    out = {"genome": host, "contig_off": np.asarray(w["contig_off"]), "cands": np.asarray(w["cands"][b0:int(w["cand_off"][c1])]),
           "cand_off": np.asarray(w["cand_off"][c0:c1 + 1]) - b0}
    if state is not None and state["found"]:
        nc, p_cf, p_ct, p_s1, p_e1, p_mn = state["found"]
        out["copy_first"] = ctx.download(p_cf, (c1 - c0) + 1, np.int32)
        out["contig"], out["start1"] = ctx.download(p_ct, nc, np.int32), ctx.download(p_s1, nc, np.int64)
        out["end1"], out["minus"] = ctx.download(p_e1, nc, np.int64), ctx.download(p_mn, nc, np.uint8)
    else:
        k0 = int(w["copy_first"][c0])
        out["copy_first"] = np.asarray(w["copy_first"][c0:c1 + 1]) - k0
        k1 = k0 + int(out["copy_first"][-1])
        out["contig"], out["start1"], out["end1"], out["minus"] = (np.asarray(w[k][k0:k1]) for k in ("contig", "start1", "end1", "minus"))
    return out


def _candidate(wv, c):
    a, b = int(wv["copy_first"][c]), int(wv["copy_first"][c + 1])
    copies = [(int(wv["contig"][i]), int(wv["start1"][i]), int(wv["end1"][i]), int(wv["minus"][i])) for i in range(a, b)]
    cand = wv["cands"][wv["cand_off"][c]:wv["cand_off"][c + 1]].tobytes().decode()
    return cand, copies


def _contigs(wv):
    co = wv["contig_off"]
    return {ci: wv["genome"][co[ci]:co[ci + 1]].tobytes() for ci in range(len(co) - 1)}


def verify(wv, calls, cons, count):
    """full-size parity spot check: the oracle chain on random candidates of THIS workload vs the GPU calls"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_pipeline as OP

    contigs = _contigs(wv)
    n_cand = len(wv["cand_off"]) - 1
    rng = np.random.default_rng(12345)
    bad = []
    info_names = {0: "", 1: "nb", 2: "fl1", 3: "EXC"}
    picks = rng.permutation(n_cand)[:count]
    n_te = 0
    for c in picks:
        cand, copies = _candidate(wv, c)
        exp = OP.fine_stage_candidate("tir", cand, copies, contigs, plant=1)
        r = calls[c]
        got = [bool(r["is_te"]), info_names[int(r["info"])],
               cons[r["cons_off"]:r["cons_off"] + r["cons_len"]].tobytes().decode() if r["is_te"] else "", int(r["row_num"])]
        n_te += bool(r["is_te"])
        if got != exp:
            bad.append(int(c))
    return {"checked": int(len(picks)), "mismatches": len(bad), "bad_candidates": bad[:10], "te_calls_in_sample": n_te,
            "against": "oracle chain (tests/oracle_pipeline.py over oracle/*.c) on the copy table of the step"}


# ---------------------------------------------------------------------------------------------
# CPU baseline: the oracle chain on the host cores, worker processes that never touch HIP (spawned, not forked)
# ---------------------------------------------------------------------------------------------
def _cpu_worker(job):
    path, genome_len, cands, budget_s = job
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_pipeline as OP

    wv = dict(np.load(path + ".npz"))
    wv["genome"] = np.memmap(path + ".genome", dtype=np.uint8, mode="r", shape=(genome_len,))
    contigs = _contigs(wv)
    t0 = time.perf_counter()
    done = 0
    for c in cands:
        cand, copies = _candidate(wv, c)
        OP.fine_stage_candidate("tir", cand, copies, contigs, plant=1)
        done += 1
        if time.perf_counter() - t0 > budget_s and done >= 4:
            break
    return done, time.perf_counter() - t0


def cpu_baseline(wv, budget_s, threads=0, with_copies=False):
    """the oracle chain (oracle/*.c through tests/oracle_pipeline.py: a CPU port of the same step -- gather + alignment +
    sparse columns + judge on the copy table of the workload) timed on a bounded sample of the same candidates on this host,
    fanned out over worker processes the way the reference fans candidates out (ProcessPoolExecutor, Util.py:8141).
    threads = 0: min(40, cores).  with_copies: the CPU twin of the copy finder (index + lookup over the whole genome, single
    thread) is timed as well and charged to the per-candidate rate."""
    import multiprocessing as mp
    import tempfile

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    threads = int(threads) if threads and threads > 0 else min(40, os.cpu_count() or 1)
    n_cand = len(wv["cand_off"]) - 1
    order = np.random.default_rng(1).permutation(n_cand)
    base = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir(), "hite_bench_%d" % os.getpid())
    try:
        np.asarray(wv["genome"]).tofile(base + ".genome")
        np.savez(base + ".npz", **{k: np.asarray(v) for k, v in wv.items() if k != "genome"})
        glen = int(len(wv["genome"]))
        t0 = time.perf_counter()
        if threads == 1:
            res = [_cpu_worker((base, glen, order, budget_s))]
        else:
            with mp.get_context("spawn").Pool(threads) as pool:
                res = pool.map(_cpu_worker, [(base, glen, order[k::threads], budget_s) for k in range(threads)])
        wall = time.perf_counter() - t0
    finally:
        for suf in (".genome", ".npz"):
            try:
                os.remove(base + suf)
            except OSError:
                pass
    done = sum(r[0] for r in res)
    busy = max(r[1] for r in res)     # the slowest worker's judging time (excludes interpreter start-up of the spawned workers)
    rate = done / busy
    note = "%d random candidates of the same workload (%.1f s judging, %.1f s wall incl. worker start-up), oracle chain, %d worker process%s" % (
        done, busy, wall, threads, "" if threads == 1 else "es")
    outd = {"value": round(rate, 3), "unit": "candidates/s", "cores": threads, "kind": "port", "sample": note}
    if with_copies:
        import oracle_lib as O

        co = wv["contig_off"]
        contigs = [wv["genome"][co[i]:co[i + 1]].tobytes() for i in range(len(co) - 1)]
        sample = order[:max(8, min(done, 512))]
        cands = [wv["cands"][wv["cand_off"][c]:wv["cand_off"][c + 1]].tobytes() for c in sample]
        import ctypes
        t0 = time.perf_counter()
        O.find_copies(contigs, cands)
        t_all = time.perf_counter() - t0
        O.lib().orc_find_copies_index_seconds.restype = ctypes.c_double
        t_index = float(O.lib().orc_find_copies_index_seconds())      # one index build, timed inside the twin
        per_cand = max(0.0, t_all - t_index) / len(cands)
        outd["copy_finding"] = {"index_s": round(t_index, 2), "lookup_s_per_candidate": round(per_cand, 5), "cores": 1,
                                "note": "CPU twin of the copy finder (oracle/hite_oracle_copies.c); the index is residency set-up on both sides"}
        # charge the lookup at the same parallel width as the judging
        outd["value"] = round(1.0 / (1.0 / rate + per_cand / threads), 3)
        outd["sample"] = note + "; copy finding (CPU twin) charged: %.2f ms per candidate per core" % (1000.0 * per_cand)
    else:
        outd["sample"] = note + "; copy finding not charged to the CPU leg (genomes above 1 Gbp: the CPU index alone takes minutes; --cpu-copies)"
    return outd


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
    mbp, tir_d, ltr_d = CONFIGS[args.config]
    if args.genome_mbp is not None:
        mbp = args.genome_mbp
    G = mbp * 1_000_000
    n_tir = args.tir_families if args.tir_families is not None else max(1, int(tir_d * mbp))
    n_ltr = args.ltr_families if args.ltr_families is not None else max(0, int(ltr_d * mbp))
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
        out = {"metric": "coarse_boundary step (stage 3.1: all-vs-all seeding + FMEA) on the %s synthetic genome" % ("1 Gbp" if mbp == 1000 else "%d Mbp" % mbp),
               "value": round(world * mbp * args.steps / elapsed, 2), "unit": "Mbp/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "u8", "data": "synthetic",
               "config": {"workload": "%s genome: %d Mbp, %d TIR + %d LTR families, 1 Mbp segments, one chunk%s" %
                                      (args.config, mbp, n_tir, n_ltr, "; one genome per GPU (replicas)" if world > 1 else ""),
                          "seeds": stats[0], "anchors": stats[1], "clusters": stats[2], "hsp_records": stats[3], "repeat_intervals": n_iv},
               "roofline": {"bound": "hbm", "achieved": round((12.0 * stats[0] * 9 + 24.0 * stats[1] * 5 + 48.0 * stats[3] * 5) / (elapsed / args.steps) / 1e9, 2),
                            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                            "note": "algorithmic bytes = radix passes x 2 x record size over seeds / anchors / HSP records (whole step, not one kernel)"}}
        out["roofline"]["frac"] = round(out["roofline"]["achieved"] / PEAK_HBM_GBS, 5)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = coarse_cpu_baseline(args, min(mbp, 20))
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


if __name__ == "__main__":
    main()
