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
  --scaling strong (default for --gpus N > 1): ONE batch is sharded over the ranks (BASELINE config 4); the line carries the weak
                   form as the side block `weak`,
  --scaling weak  (default for one GPU): every rank judges its own candidate batch (N x the work),
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
    # name: (genome Mbp, TIR families per Mbp, LTR families per Mbp)  -- BASELINE.json configs[1] / configs[2] / configs[4]
    "C2": (100, 5.0, 0.0),
    "C3": (1000, 2.5, 2.5),
    "C5": (300, 2.5, 2.5),     # panHiTE: one 300 Mbp genome per GPU, 70 % of the families shared, libraries merged (c5_mode)
    # configs[3] as ONE GPU sees it: the C3 genome and candidate batch, of which this process judges rank 0's strong-scaling share
    # of 8 (6 250 candidates, length-balanced as hite_amd.dist deals them) -- small-batch efficiency measurable without a node
    "C4share": (1000, 2.5, 2.5),
}


def valu_peak():
    """wave64 issue rates MEASURED on MI355X by tools/valu_issue_bench.hip (profiles/rNN_valu_issue.txt, newest), at 8 waves per SIMD, as two
    classes and no blend: instructions that issue every ~4.1 cycles per SIMD (3-operand, bit-field, carry, DPP, shifts: what the
    alignment recurrence is mostly made of) and those that issue every ~2.1 cycles (2-operand add / xor).  -> (G wave-inst/s of
    the 4-cycle class, of the 2-cycle class): medians of the measured lines of each class."""
    c4, c2 = [], []
    try:
        import glob
        for line in open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_valu_issue.txt")))[-1]):
            f = line.split()
            if len(f) < 4 or line.startswith("#"):
                continue
            try:
                waves, rate, cyc = int(f[-3]), float(f[-2]), float(f[-1])
            except ValueError:
                continue
            if waves != 8:
                continue
            if cyc < 3.0:
                c2.append(rate)
            elif cyc < 6.0:
                c4.append(rate)       # (v_cndmask at 23 cycles is VCC-serialised in that micro-benchmark: neither class)
    except OSError:
        pass
    med = lambda v, d: sorted(v)[len(v) // 2] if v else d  # noqa: E731
    return med(c4, 600.0), med(c2, 1180.0)


def _kept_profile(stem):
    """the newest committed profile file profiles/rNN_<stem> (rounds keep their own files) -> (parsed JSON, its name)"""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + stem)), reverse=True):
        try:
            return json.load(open(path)), "profiles/" + os.path.basename(path)
        except (OSError, ValueError):
            continue
    return {}, None


def kept_counters():
    """per-kernel SQ counters of the committed profile (profiles/rNN_sq_counters.json: rocprofv3 --pmc SQ_INSTS_VALU ...)"""
    return _kept_profile("sq_counters.json")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_devices(torch, dist, rank, local_rank, world):
    """-> (backend, gpu index of this rank, its device, the device a collective's tensors live on); starts the process group.
    HITE_BENCH_BACKEND=gloo: a FUNCTIONAL run of the N > 1 paths on a box with fewer GPUs than ranks (the ranks share the GPUs, the
    collectives move host copies): what a one-GPU box can check of the multi-GPU lines before a node runs them.  Not a measurement
    -- such a line says so in `data`."""
    backend = os.environ.get("HITE_BENCH_BACKEND", "nccl")
    gpu_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    cdev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return backend, gpu_index, dev, cdev


def data_note(torch, backend):
    return "synthetic" if backend == "nccl" else ("synthetic; FUNCTIONAL run over %s with the ranks sharing %d GPU(s): not a measurement" %
                                                  (backend, torch.cuda.device_count()))


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
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="default: strong when --gpus > 1 (BASELINE config 4: ONE 1 Gbp batch sharded over the GPUs; the line then carries the "
                         "weak form -- every rank the whole batch -- as the side block `weak`), weak otherwise")
    ap.add_argument("--genomes", type=int, default=1,
                    help="config C5 on ONE process: this many population genomes one after another on the GPU, then the merge of all their libraries (8 = the configured size)")
    ap.add_argument("--share-of", type=int, default=0,
                    help="one process: judge only rank 0's strong-scaling share of this many ranks (config C4share: 8)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="worker processes of the cpu_baseline leg (0 = min(40, cores of this host))")
    ap.add_argument("--cpu-copies", action="store_true",
                    help="cpu_baseline: also time the CPU twin of the copy finder on the whole genome (default up to 1 Gbp, where its index takes about a minute and a half)")
    ap.add_argument("--copies", choices=["found", "truth"], default="found",
                    help="found: copy finding (minimizer index lookup) runs inside the timed step; truth: the generator's copy table is the input")
    ap.add_argument("--verify", type=int, default=512, help="candidates re-judged with the CPU oracle chain after the timed region (0 = none)")
    ap.add_argument("--no-coarse", action="store_true", help="skip the coarse-stage block of the default line")
    ap.add_argument("--no-modes", action="store_true", help="skip the block that re-runs the step in the other interval mode of the copy finder")
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
    if args.config == "C5":
        return c5_mode(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    backend, gpu_index, dev, cdev = rank_devices(torch, dist, rank, local_rank, world)

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
    if args.scaling is None:
        args.scaling = "strong" if world > 1 else "weak"
    strong = args.scaling == "strong" and world > 1
    share_of = args.share_of if args.share_of > 1 else (8 if args.config == "C4share" else 0)
    if world > 1:
        share_of = 0
    t0 = time.time()
    # same genome on every rank (replicated); weak scaling: rank-specific candidate draw, strong scaling: one draw, sharded
    w = synth.make_workload(genome_bp=G, n_tir=n_tir, n_ltr=n_ltr, cands_per_family=args.cands_per_family, seed=args.seed,
                            device=dev, cand_seed=args.seed + 7919 + (0 if strong else 104729 * rank))
    setup_s = time.time() - t0
    n_all = len(w["cand_off"]) - 1
    shares = None
    if strong or share_of:
        # ONE batch over the ranks: length-balanced block-cyclic shares (hite_amd.dist: the step time of a rank is set by its longest
        # alignments), the share as its own CSR; the all-gathered records go back to candidate order by the inverse permutation
        ids, shares = hd.shard_candidates_balanced(w["cand_off"], w["copy_first"], rank if strong else 0, world if strong else share_of)
        if not strong:
            shares = None
        L = {}
        L["cands"], L["cand_off"] = hd.gather_csr(w["cands"], w["cand_off"], ids)
        cf64 = np.asarray(w["copy_first"], dtype=np.int64)
        for k_ in ("contig", "start1", "end1", "minus"):
            L[k_], new_cf = hd.gather_csr(w[k_], cf64, ids)
        L["copy_first"] = new_cf.astype(np.int32)
    else:
        L = {k_: w[k_] for k_ in ("cands", "cand_off", "copy_first", "contig", "start1", "end1", "minus")}
    n_cand = len(L["cand_off"]) - 1
    c0, c1, b0, b1, k0, k1 = 0, n_cand, 0, int(L["cand_off"][-1]), 0, len(L["contig"])

    ctx = hite_amd.Context(gpu_index)
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

    def device_batch(Lx):
        """the device-resident inputs and outputs of one candidate batch (CSR of candidates, copy table of the generator for --copies truth)"""
        nc_ = len(Lx["cand_off"]) - 1
        bases = int(Lx["cand_off"][-1])
        Bx = {"n_cand": nc_, "bases": bases, "n_truth": len(Lx["contig"]), "cons_cap": bases + 200 * nc_ + 4096}
        Bx["calls"] = torch.zeros(max(1, nc_) * 32, dtype=torch.uint8, device=dev)
        Bx["cons"] = torch.zeros(Bx["cons_cap"] + 64, dtype=torch.uint8, device=dev)
        Bx["cand"] = up(np.concatenate([Lx["cands"], np.zeros(64, np.uint8)]))
        Bx["cand_off"] = up(np.asarray(Lx["cand_off"], dtype=np.int64))
        Bx["cf"] = up(np.asarray(Lx["copy_first"], dtype=np.int32))
        Bx["ct"], Bx["s1"], Bx["e1"] = up(Lx["contig"]), up(Lx["start1"]), up(Lx["end1"])
        Bx["mn"] = up(np.concatenate([Lx["minus"], np.zeros(16, np.uint8)]))
        return Bx

    B = device_batch(L)
    d_calls, d_cons, cons_cap = B["calls"], B["cons"], B["cons_cap"]
    state = {"found": None, "n_copies": k1 - k0}
    n_merge = n_all if strong else world * n_cand

    def run_batch(Bx):
        """one pass of the hot path over one device batch -> the pipeline's statistics"""
        if Bx["n_cand"] <= 0:
            return np.zeros(12, dtype=np.int64)
        if args.copies == "found":
            nc, p_cf, p_ct, p_s1, p_e1, p_mn, _p_an = ctx.find_copies_dev(Bx["n_cand"], Bx["cand"].data_ptr(), Bx["cand_off"].data_ptr(), Bx["bases"], sp)
            p_cl = ctx.copy_clips_dev()      # (the rows are padded by the clipped bases; zero words with HITE_COPY_INTERVAL=whole)
            state["found"] = (nc, p_cf, p_ct, p_s1, p_e1, p_mn, p_cl)
            state["n_copies"] = nc
            return ctx.flank_region_align_dev("tir", 1, Bx["n_cand"], Bx["cand"].data_ptr(), Bx["cand_off"].data_ptr(), p_cf, nc, p_ct, p_s1, p_e1, p_mn,
                                              50, Bx["calls"].data_ptr(), Bx["cons"].data_ptr(), Bx["cons_cap"], sp, d_clip=p_cl)
        return ctx.flank_region_align_dev("tir", 1, Bx["n_cand"], Bx["cand"].data_ptr(), Bx["cand_off"].data_ptr(), Bx["cf"].data_ptr(), Bx["n_truth"],
                                          Bx["ct"].data_ptr(), Bx["s1"].data_ptr(), Bx["e1"].data_ptr(), Bx["mn"].data_ptr(), 50, Bx["calls"].data_ptr(),
                                          Bx["cons"].data_ptr(), Bx["cons_cap"], sp)

    def step():
        st = run_batch(B)
        stream.synchronize()
        merged = None
        if world > 1:   # merge the boundary calls (RCCL over xGMI): ONE all-gather of the 32-byte records
            mine = d_calls[: n_cand * 32].to(cdev)
            merged = (hd.allgather_calls_balanced(mine, shares, alias=True) if shares is not None
                      else hd.allgather_calls(mine, n_merge, alias=True))     # (a view of the merge buffer: read before the next step)
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
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    calls = d_calls.cpu().numpy().view(CALL_DTYPE)[:n_cand].copy()
    n_te = int((calls["is_te"] != 0).sum())
    if world > 1:
        tn = torch.tensor([n_te], dtype=torch.int64, device=cdev)
        dist.all_reduce(tn)
        n_te_all = int(tn.item())
        merged_te = int((merged.cpu().numpy().view(CALL_DTYPE)["is_te"] != 0).sum())
        assert merged_te == n_te_all, "all-gathered records disagree with the per-rank counts"
    else:
        n_te_all = n_te

    weak_blk = None
    if strong:
        # The WEAK form as a side block of a strong-scaling line: every rank judges the whole batch (per-GPU work fixed as N grows),
        # a few steps outside the headline's timed region, without the merge of the calls.  No collective sits inside the try
        # blocks: a rank that fails still reaches the barrier and the reductions below.
        ok_w, err_w, t_w, k_w = 1, "", 0.0, max(1, min(3, args.steps))
        Bw = None
        try:
            Bw = device_batch({k_: w[k_] for k_ in ("cands", "cand_off", "copy_first", "contig", "start1", "end1", "minus")})
            for _ in range(2):            # (the arenas grow to this batch's size, then are consolidated)
                run_batch(Bw)
                stream.synchronize()
            torch.cuda.synchronize()
        except Exception as e:
            ok_w, err_w = 0, "%s: %s" % (type(e).__name__, e)
        dist.barrier()
        t_w0 = time.perf_counter()
        if ok_w:
            try:
                for _ in range(k_w):
                    run_batch(Bw)
                    stream.synchronize()
                torch.cuda.synchronize()
            except Exception as e:
                ok_w, err_w = 0, "%s: %s" % (type(e).__name__, e)
        t_w = time.perf_counter() - t_w0
        tw = torch.tensor([t_w, -float(ok_w)], dtype=torch.float64, device=cdev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        if float(tw[1].item()) == -1.0:
            t_w = float(tw[0].item())
            weak_blk = {"scaling": "weak", "value": round(world * n_all * k_w / t_w, 2), "unit": "candidates/s", "steps": k_w,
                        "ms_per_step": round(1000.0 * t_w / k_w, 3), "candidates_per_gpu": n_all,
                        "note": "every rank judges the whole %d-candidate batch; max over ranks between barriers; the all-gather of the calls is not in this block" % n_all}
        else:
            weak_blk = {"scaling": "weak", "error": err_w or "another rank failed"}
        del Bw
        step()                            # the copy table and the calls on the device are the strong share's again
        torch.cuda.synchronize()

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
            alg["radix_sort_hits"] = float(cs[1]) * 16.0                    # 16 B per hit, whatever the number of passes (passes = the sort's own business)
            _ = passes
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
            # profiles/rNN_pmc_traffic.json, the newest round's); a profiler stage can cover several kernel instantiations
            traffic = None
            try:
                pmc, pmc_src = _kept_profile("pmc_traffic.json")
                names = {"align_fwd4": ["align_fwd4_kernel", "align_fwd_kernel<4>"], "align_tb": ["align_tb_kernel"],
                         "align_fwd_wide": ["align_fwd_kernel<8>", "align_fwd_kernel<16>", "align_fwd_kernel<32>"],
                         "radix_sort_hits": ["rs_scatter_staged_kernel", "rs_hist_kernel<10, 32>"],
                         "judge_kernel": ["jblk::judge_kernel", "jwav::judge_wave_kernel"]}.get(dom, [dom])
                tot = sum(pmc[k_]["bytes_per_launch"] * pmc[k_]["launches"] for k_ in names if k_ in pmc)
                runs = max([pmc.get(k_, {}).get("launches", 0) for k_ in names] + [0])
                traffic = int(tot / runs) if tot and runs else None      # per launch of the stage
            except Exception:
                traffic = None
            roof = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 3), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(achieved / PEAK_HBM_GBS, 6), "traffic": traffic,
                    "avg_launch_ms": round(avg_launch_ms, 4), "alg_bytes_per_launch": int(bytes_per_launch)}
            if dom in ("align_fwd4", "align_tb", "align_fwd_wide"):
                # An alignment kernel is bound by vector-instruction ISSUE (its wavefronts' dependent chains), not by HBM: the block
                # leads with that -- wave-level vector instructions of THIS kernel (committed SQ counters of the same workload, scaled
                # by this run's pair-columns) / its live launch time, against the measured issue rate of the 4-cycle class -- and keeps
                # the HBM view beside it on the IRREDUCIBLE bytes (SURVEY 8(d) gives the aligner no byte formula: 1 B of row base per
                # pair-column forward; the ops, 2 B per centre position, for the traceback); the check points / boundary records
                # (4 B per pair-column) are the kernel's own traffic, listed as such, next to the counter traffic.
                slow_, _fast = valu_peak()
                cnt_file_, cnt_src_ = kept_counters()
                e_ = cnt_file_.get({"align_fwd_wide": "align_fwd_wide8"}.get(dom, dom), {})
                cols_file_ = cnt_file_.get("_columns_per_step", 0.0)
                irreducible = {"align_fwd4": 1.0 * cols_step, "align_fwd_wide": 1.0 * cols_step * (per_step["wide"] / pairs_step),
                               "align_tb": ops_bytes}[dom] * steps / max(1, cnt)
                hbm_view = {"bound": "hbm", "achieved": round(irreducible / (avg_launch_ms * 1e-3) / 1e9, 3), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": round(irreducible / (avg_launch_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 6), "alg_bytes_per_launch": int(irreducible),
                            "note": "irreducible bytes: 1 B row base per pair-column (forward) / the ops (traceback)",
                            "with_own_records": {"alg_bytes_per_launch": int(bytes_per_launch), "achieved": round(achieved, 3),
                                                 "frac": round(achieved / PEAK_HBM_GBS, 6),
                                                 "note": "+ 4 B per pair-column of check points / boundary records: the kernel's own traffic, not the algorithm's"},
                            "traffic": traffic}
                if e_.get("valu_inst_per_step") and cols_file_ > 0 and slow_ > 0:
                    inst_launch = e_["valu_inst_per_step"] * (cols_step / cols_file_) * steps / max(1, cnt)
                    rate_ = inst_launch / (avg_launch_ms * 1e-3) / 1e9
                    roof = {"kernel": dom, "bound": "valu_issue", "achieved": round(rate_, 1), "peak": round(slow_, 1), "unit": "G wave64-inst/s",
                            "frac": round(rate_ / slow_, 4), "traffic": traffic, "avg_launch_ms": round(avg_launch_ms, 4),
                            "wave_inst_per_launch": int(inst_launch), "wave_inst_per_pair_column": round(e_["valu_inst_per_step"] / cols_file_, 4),
                            "peak_note": "measured issue rate of the 4-cycle instruction class at 8 waves per SIMD (tools/valu_issue_bench.hip); "
                                         "against the 2-cycle class the fraction is %.4f" % (rate_ / _fast if _fast > 0 else 0.0),
                            "source": cnt_src_, "hbm": hbm_view}
                else:
                    roof["hbm_irreducible"] = hbm_view
            # the alignment kernels are bound by vector-instruction issue, not by HBM: report the whole group against the MEASURED
            # issue rate (tools/valu_issue_bench.hip -> profiles/r02_valu_issue.txt); instructions per column from the kept counters
            align_ms = sum(ms for k, (ms, _c) in merged_prof.items() if k.startswith("align_")) / steps
            if align_ms > 0 and cols_step > 0:
                slow, fast = valu_peak()
                cnt_file, cnt_src = kept_counters()
                inst = sum(v.get("valu_inst_per_step", 0.0) for k, v in cnt_file.items() if k.startswith("align_"))
                cols_file = cnt_file.get("_columns_per_step", 0.0)
                blk = {"ms_per_step": round(align_ms, 3), "pairs_per_step": int(per_step["pairs"]), "columns_per_step": int(cols_step),
                       "band_gcells_per_s": round(128.0 * cols_step / (align_ms * 1e-3) / 1e9, 1),
                       "certified_frac": round(per_step["certified"] / pairs_step, 4), "wide_frac": round(per_step["wide"] / pairs_step, 4),
                       "fallback_per_step": per_step["fallback"], "dropped_per_step": per_step["dropped"], "exact_cap": per_step["exact_cap"],
                       "peak_ginst_4cycle_class": round(slow, 1), "peak_ginst_2cycle_class": round(fast, 1)}
                if inst > 0 and cols_file > 0:
                    per_col = inst / cols_file            # wave-level instructions per pair-column
                    rate = per_col * cols_step / (align_ms * 1e-3) / 1e9
                    # no blended peak: the fraction against the 4-cycle class (an upper bound of the true fraction: some of the mix
                    # issues in 2 cycles) and against the 2-cycle class (a lower bound)
                    blk["valu_issue"] = {"achieved": round(rate, 1), "unit": "G wave64-inst/s", "peak_4cycle_class": round(slow, 1),
                                         "peak_2cycle_class": round(fast, 1), "frac_of_4cycle_peak": round(rate / slow, 4),
                                         "frac_of_2cycle_peak": round(rate / fast, 4),
                                         "wave_inst_per_pair_column": round(per_col, 4), "source": cnt_src}
                roof["align"] = blk
            # the whole step by SURVEY.md 8(d)'s formulas, verbatim: B_copy = Q/4 + 12 M_q + 16 hits; B_gather = sum R_c (W_c/4 + W_c);
            # B_vote = sum (R_c W_c + 20 W_c)  -- the path as built is instruction / latency bound, not bandwidth bound
            if args.copies == "found":
                cs_ = [int(x) for x in ctx.copy_stats()]
                b_copy = (b1 - b0) / 4.0 + 12.0 * cs_[0] + 16.0 * cs_[1]
            else:
                b_copy = 0.0
            b_gather = 1.25 * float(stats[1] + stats[5])
            b_vote = float(stats[2] + stats[6]) + 20.0 * float(stats[11])
            sb = b_copy + b_gather + b_vote
            roof["step"] = {"survey_alg_bytes": int(sb), "copy": int(b_copy), "gather": int(b_gather), "vote": int(b_vote),
                            "GB/s": round(sb / (ms_per_step * 1e-3) / 1e9, 1), "frac": round(sb / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 5),
                            "note": "SURVEY 8(d) formulas over the whole step time; the step is instruction / latency bound (see align.valu_issue, kernels)"}
            # every stage with an algorithmic byte count, against the same HBM peak (what each is really bound by: DESIGN.md section 4)
            roof["stages"] = {k: {"ms_per_step": round(ms / steps, 3), "achieved": round(alg[k] / (ms / steps * 1e-3) / 1e9, 1),
                                  "frac": round(alg[k] / (ms / steps * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
                              for k, (ms, _c) in sorted(merged_prof.items()) if k in alg and ms > 0}
        out = {
            "metric": "candidate TE boundaries/sec on %s synthetic genome (fine stage: copy finding+gather+align+vote+judge)" % ("1 Gbp" if mbp == 1000 else "%d Mbp" % mbp),
            "value": round(value, 2), "unit": "candidates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u8", "data": data_note(torch, backend),
            "config": {"workload": "%s: %d Mbp synthetic genome, %d TIR + %d LTR families, %d candidates%s judged as TIR (%s)" %
                                   (args.config if args.genome_mbp is None else "custom", mbp, n_tir, n_ltr, total_cands if strong else n_cand,
                                    " sharded over %d GPUs" % world if strong else
                                    (" = rank 0's length-balanced share of %d ranks of the %d-candidate batch" % (share_of, n_all) if share_of else "/GPU"),
                                    "copy finding by minimizer-index lookup inside the timed step; index build %.1f s untimed" % index_s
                                    if args.copies == "found" else "copy table = generator truth; copy finding not in the timed path"),
                       "genome_bp": G, "candidates_per_gpu": n_cand, "us_per_candidate": round(1000.0 * ms_per_step / max(1, n_cand), 4),
                       "candidate_bases": b1 - b0, "copies": int(state["n_copies"]),
                       "copy_table": args.copies, "rows_aligned_per_step": rows,
                       "pipeline_stats": [int(x) for x in stats], "copy_stats": [int(x) for x in ctx.copy_stats_ext()] if args.copies == "found" else None,
                       "align_stats_per_step": {k_: int(v_) for k_, v_ in per_step.items()},
                       "is_te": n_te_all, "parallelism": "replicated genome, candidates %s x%d, all-gather of 32-B calls" %
                                                        ("sharded (length-balanced block-cyclic)" if strong else "per rank", world),
                       "setup_s": round(setup_s, 1)},
            "roofline": roof,
            "kernels": {k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in sorted(kern.items())},
        }
        if weak_blk is not None:
            out["weak"] = weak_blk
        if hasattr(ctx.lib, "hite_debug_judge_clocks"):   # only in a -DJUDGE_CLOCKS development build
            import ctypes
            buf = (ctypes.c_ulonglong * 16)()
            ctx.lib.hite_debug_judge_clocks(buf, 1)
            out["judge_phase_ticks"] = [int(x) for x in buf[:12]]
        # the host view of the step (candidates, the copy table the GPU used, its calls) is taken BEFORE the coarse block: that one
        # re-packs the genome and rebuilds the index the copy table lives in
        wv = None
        if (world == 1 and not args.no_cpu_baseline) or args.verify > 0:
            wv = host_workload(dict(w, **L), ctx, state if args.copies == "found" else None, c0, c1)
        if world == 1 and args.copies == "found" and not args.no_modes and n_cand > 0:
            # the same step in the OTHER interval mode of the copy finder, outside the headline's timed region (same batch, same
            # kernels; a sample re-judged by the oracle chain on that copy table).  The default (and the headline): copy records in the
            # REFERENCE'S coordinates -- reference_start + 1 .. reference_end, Util.py:8026 -- with the rows padded by the clipped
            # candidate bases (hite_flank_region_align_clip_dev); the other: the whole-candidate intervals of rounds 2-4
            whole_now = os.environ.get("HITE_COPY_INTERVAL", "") in ("whole", "0")
            names = {True: "whole candidate (aligned part + the clipped ends on its diagonal; HITE_COPY_INTERVAL=whole, the default of rounds 2-4)",
                     False: "aligned part of the candidate, as get_copies_minimap2 reports it (Util.py:8026); rows padded by the clipped bases"}
            try:
                ctx.copy_config(whole_now)          # (True = aligned: the other mode of a run that was started in the whole-candidate mode)
                step()
                ctx.align_stats(reset=True)
                torch.cuda.synchronize()
                k_m = max(1, min(3, args.steps))
                t_m = time.perf_counter()
                for _ in range(k_m):
                    step()
                torch.cuda.synchronize()
                ms_m = 1000.0 * (time.perf_counter() - t_m) / k_m
                st_m = ctx.align_stats()
                calls_m = d_calls.cpu().numpy().view(CALL_DTYPE)[:n_cand].copy()
                blk = {"this_run": {"interval": names[whole_now], "ms_per_step": round(ms_per_step, 3), "is_te": n_te_all,
                                    "wide_fallback_per_step": per_step["fallback"]},
                       "other_mode": {"interval": names[not whole_now], "ms_per_step": round(ms_m, 3), "value": round(n_cand / (ms_m * 1e-3), 2),
                                      "unit": "candidates/s", "steps": k_m, "copies": int(state["n_copies"]),
                                      "is_te": int((calls_m["is_te"] != 0).sum()), "wide_fallback_per_step": st_m["fallback"] / k_m,
                                      "dropped_per_step": st_m["dropped"] / k_m,
                                      "certified_frac": round(st_m["certified"] / max(1, st_m["pairs"]), 4)}}
                if args.verify > 0:
                    wv_m = host_workload(dict(w, **L), ctx, state, c0, c1)
                    v_m = verify(wv_m, calls_m, d_cons.cpu().numpy(), min(args.verify, 128))
                    blk["other_mode"]["verify"] = {k_: v_m[k_] for k_ in ("checked", "mismatches", "bad_candidates", "te_calls_in_sample") if k_ in v_m}
                out["copy_interval_modes"] = blk
            except Exception as e:
                out["copy_interval_modes"] = {"error": "%s: %s" % (type(e).__name__, e)}
            finally:
                ctx.copy_config(None)
                step()                      # the device copy table and calls are this run's mode's again
                torch.cuda.synchronize()
        if world == 1 and not args.no_coarse:
            # north_star's >= 20x target is phrased on the coarse_boundary step: measured here, after the headline's timed region, on
            # the same resident genome (stage 3.1: index + all-vs-all seeding + FMEA over the whole genome as one chunk)
            try:
                out["coarse"] = coarse_block(ctx, args, mbp, n_tir, n_ltr, torch, not args.no_cpu_baseline, w=dict(w, **L))
            except Exception as e:
                out["coarse"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(wv, args.cpu_seconds, args.cpu_threads, args.cpu_copies or mbp <= 1000)
            except Exception as e:   # the GPU line must not depend on the CPU leg
                out["cpu_baseline"] = {"value": None, "unit": "candidates/s", "cores": args.cpu_threads, "kind": "port",
                                       "sample": "failed: %s: %s" % (type(e).__name__, e)}
        if args.verify > 0:
            # the oracle chain re-judges on the SAME copy table the GPU used; the copy table itself is compared with the CPU twin's
            # on the sample of the cpu_baseline leg (copy_tables); outside the timed region
            # (with several ranks the others wait at the closing barrier while rank 0 re-judges its sample: a smaller one there)
            out["verify"] = verify(wv, calls, d_cons.cpu().numpy(), args.verify if world == 1 else min(args.verify, 64))
            ct = (out.get("cpu_baseline") or {}).pop("copy_tables", None)
            if ct is not None:
                out["verify"]["copy_tables"] = ct
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
        nc, p_cf, p_ct, p_s1, p_e1, p_mn, p_cl = state["found"]
        out["copy_first"] = ctx.download(p_cf, (c1 - c0) + 1, np.int32)
        out["contig"], out["start1"] = ctx.download(p_ct, nc, np.int32), ctx.download(p_s1, nc, np.int64)
        out["end1"], out["minus"] = ctx.download(p_e1, nc, np.int64), ctx.download(p_mn, nc, np.uint8)
        if p_cl and nc > 0:
            out["clip"] = ctx.download(p_cl, nc, np.uint32)
    else:
        k0 = int(w["copy_first"][c0])
        out["copy_first"] = np.asarray(w["copy_first"][c0:c1 + 1]) - k0
        k1 = k0 + int(out["copy_first"][-1])
        out["contig"], out["start1"], out["end1"], out["minus"] = (np.asarray(w[k][k0:k1]) for k in ("contig", "start1", "end1", "minus"))
    return out


def _candidate(wv, c):
    a, b = int(wv["copy_first"][c]), int(wv["copy_first"][c + 1])
    cl = wv.get("clip")
    copies = [(int(wv["contig"][i]), int(wv["start1"][i]), int(wv["end1"][i]), int(wv["minus"][i]), 0, int(cl[i]) if cl is not None else 0) for i in range(a, b)]
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
    anchors = {"none": 0, "exact": 0, "interior": 0, "end": 0}     # per judged pass of the sample (tests/oracle_pipeline.py anchor_class)
    for c in picks:
        cand, copies = _candidate(wv, c)
        msas = []
        exp = OP.fine_stage_candidate("tir", cand, copies, contigs, plant=1, keep_msa=msas)
        for m_ in msas:
            anchors[OP.anchor_class(cand, m_)] += 1
        r = calls[c]
        got = [bool(r["is_te"]), info_names[int(r["info"])],
               cons[r["cons_off"]:r["cons_off"] + r["cons_len"]].tobytes().decode() if r["is_te"] else "", int(r["row_num"])]
        n_te += bool(r["is_te"])
        if got != exp:
            bad.append(int(c))
    return {"checked": int(len(picks)), "mismatches": len(bad), "bad_candidates": bad[:10], "te_calls_in_sample": n_te,
            "against": "oracle chain (tests/oracle_pipeline.py over oracle/*.c) on the copy table of the step",
            "anchor_matches": dict(anchors, note="judged alignments of the sample by how the two 20-base anchors match their first common row: "
                                                 "'end' = an edit on the first / last base of a match, the only class where the real fuzzysearch "
                                                 "package could choose another start / end than the definition the goldens use (oracle/stubs.py)")}


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
    try:   # how the C port relates to the reference's own Python on the functions HiTE owns (measured in the build container)
        rt = json.load(open(os.path.join(ROOT, "profiles", "r01_reference_python_timing.json")))
        outd["reference_python_vs_c_port"] = {
            "judge_candidates_per_s_per_core": {"reference_python": rt["judge"]["python_1proc_cand_per_s"], "c_port": rt["judge"]["c_port_1thread_cand_per_s"]},
            "fmea_hsp_per_s_per_core": {"reference_python": rt["fmea"]["python_1proc_hsp_per_s"], "c_port": rt["fmea"]["c_port_1thread_hsp_per_s"]},
            "note": "the reference's Python is ~60x (judges) / ~27x (FMEA) slower per core than this C port; its blastn / minimap2 / mafft share cannot be timed here",
            "source": "profiles/r01_reference_python_timing.json"}
    except (OSError, KeyError, ValueError):
        pass
    if with_copies:
        import oracle_lib as O

        co = wv["contig_off"]
        contigs = [wv["genome"][co[i]:co[i + 1]].tobytes() for i in range(len(co) - 1)]
        sample = order[:max(8, min(done, 512))]
        cands = [wv["cands"][wv["cand_off"][c]:wv["cand_off"][c + 1]].tobytes() for c in sample]
        import ctypes
        t0 = time.perf_counter()
        twin_tab = O.find_copies(contigs, cands)
        t_all = time.perf_counter() - t0
        # the twin's copy rows for the sample against the rows the GPU step produced for the same candidates (same order)
        mism = []
        if "copy_first" in wv:
            for c, rows in zip(sample, twin_tab):
                a, b = int(wv["copy_first"][c]), int(wv["copy_first"][c + 1])
                gpu_rows = [(int(wv["contig"][i]), int(wv["start1"][i]), int(wv["end1"][i]), int(wv["minus"][i])) for i in range(a, b)]
                if gpu_rows != [r[:4] for r in rows]:
                    mism.append(int(c))
            outd["copy_tables"] = {"checked": int(len(sample)), "mismatches": len(mism), "bad_candidates": mism[:10],
                                   "copies_in_sample": int(sum(len(r) for r in twin_tab)),
                                   "against": "CPU twin of the copy finder (oracle/hite_oracle_copies.c) on the whole genome"}
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


def _prev_te_library(w, seed, n_prev):
    """a TE library "found in the chunks before" (what --prev_TE holds from chunk 2 on, main.py:496-505): the first copy of
    n_prev random families, as planted in this genome"""
    p = w["planted"]
    g = w["genome"]
    host = g.cpu().numpy() if hasattr(g, "cpu") else np.asarray(g)
    co = np.asarray(w["contig_off"])
    rng = np.random.default_rng(seed)
    fams = np.unique(p["family"][p["full"]])
    pick = set(int(f) for f in rng.permutation(fams)[:n_prev])
    out, seen = [], set()
    for i in np.flatnonzero(p["full"]):
        f = int(p["family"][i])
        if f in pick and f not in seen and 80 <= int(p["length"][i]) <= 30000:
            seen.add(f)
            a_ = int(co[int(p["contig"][i])] + p["start"][i])
            s_ = host[a_:a_ + int(p["length"][i])].tobytes()
            out.append(s_)
    return out


def _coarse_cpu_worker(job):
    """CPU leg of stage 3.1 end to end on one sub-genome: the twins of every stage of the GPU step, in its order"""
    seed, mbp = job
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C

    import oracle_lib as O
    from hite_amd import synth

    w = synth.make_workload(genome_bp=mbp * 1_000_000, n_tir=max(1, int(2.5 * mbp)), n_ltr=int(2.5 * mbp), cands_per_family=1, seed=seed)
    co = np.asarray(w["contig_off"])
    genome = np.asarray(w["genome"]).copy()
    prev = _prev_te_library(w, seed, max(1, int(0.5 * mbp)))
    t = {}
    t0 = time.perf_counter()
    # filter_tandem_repeats (Util.py:4672): the twin of the tandem masker
    L = O.lib()
    L.orc_tr_mask.restype = C.c_int64
    mask = np.zeros(len(genome), dtype=np.uint8)
    L.orc_tr_mask(genome.ctypes.data_as(O.u8p), co.ctypes.data_as(O.i64p), len(co) - 1, 500, mask.ctypes.data_as(O.u8p))
    genome[mask != 0] = ord("N")
    t["tandem"] = time.perf_counter() - t0
    # mask_genome_intactTE (Util.py:6389): copies of the TEs found so far -> N
    t1 = time.perf_counter()
    contigs = [genome[co[i]:co[i + 1]].tobytes() for i in range(len(co) - 1)]
    tab = O.find_copies(contigs, prev)
    for q, copies in zip(prev, tab):
        for (c, s1, e1, _m, _a) in copies:
            if e1 - s1 + 1 >= 0.95 * len(q):
                genome[co[c] + max(0, s1 - 1):co[c] + e1] = ord("N")
    t["prev_te"] = time.perf_counter() - t1
    # process_blast_alignments + get_longest_repeats_v4 (Util.py:4724, 4122)
    t2 = time.perf_counter()
    contigs = [genome[co[i]:co[i + 1]].tobytes() for i in range(len(co) - 1)]
    h = O.seed_allvsall(contigs, seg_len=1_000_000)
    h["chrom_names"] = ["c%d" % i for i in range(len(contigs))]
    names = O.fmea(h, 4000, 30000)
    t["search_fmea"] = time.perf_counter() - t2
    # generate_final_result + flanking_seq (Util.py:4783, 4614): the sequences of the intervals with 50 flanking bases
    t3 = time.perf_counter()
    raw = np.asarray(w["genome"])
    nbytes = 0
    for nm in names:
        c, pos = nm.split(":")
        a_, b_ = (int(x) for x in pos.split("-"))
        ci = int(c[1:])
        clen = int(co[ci + 1] - co[ci])
        s1, e1 = a_ + 1, b_
        if s1 - 1 - 50 < 0:
            s1 = 51
        if e1 + 50 > clen:
            e1 = clen - 50
        nbytes += len(raw[co[ci] + s1 - 1 - 50:co[ci] + e1 + 50].tobytes())
    t["flank"] = time.perf_counter() - t3
    return time.perf_counter() - t0, len(h["qseg"]), len(names), t, int(mask.sum())


COARSE_ALG_NOTE = ("algorithmic bytes = what the step's sorts have to move: index entries 12 B x 2 x 4 passes (hash sort carrying the position rank), "
                   "anchors 8 B x 2 x 3 passes (packed records, genomes <= 2^30 bases), HSP records 48 B x 2 x 5 (emit, 4-pass sort, gather); "
                   "whole step, not one kernel (rounds 2-4 counted 9 / 5 passes of 12- / 24-byte records)")


def coarse_alg_bytes(st):
    """st = (seeds, anchors, clusters, HSP records) of one step"""
    return 12.0 * 2 * 4 * st[0] + 8.0 * 2 * 3 * st[1] + 48.0 * 2 * 5 * st[3]


def coarse_block(ctx, args, mbp, n_tir, n_ltr, torch, with_cpu, w=None):
    """stage 3.1 END TO END on the genome of the headline workload (coarse_boundary.py:14-32 -> determine_repeat_boundary_v5,
    Util.py:4637-4670, + flanking_seq :4614), the whole genome as one chunk, everything device-resident.  A step =
      pack the chunk (read_fasta's place)  ->  tandem repeats to N (filter_tandem_repeats :4672; hite_tr_mask)
      ->  full-length copies of the TEs found so far to N (mask_genome_intactTE :6389; minimizer index + hite_find_copies + hite_genome_mask)
      ->  minimizer index of the masked chunk + all-vs-all seeding + FMEA (process_blast_alignments :4724, get_longest_repeats_v4 :4122)
      ->  the intervals with 50 flanking bases gathered from the packed genome (generate_final_result :4783, flanking_seq :4614).
    `inner` keeps the number of the earlier rounds (index + seeding + FMEA alone).  CPU leg: the twins of the same stages in the
    same order on min(40, cores) workers, one 20 Mbp sub-genome of the same family density each (the all-vs-all stage is
    super-linear in the genome, so the CPU's Mbp/s on 20 Mbp pieces flatters it against the 1 Gbp step)."""
    sc, so = ctx.seed_segments(1_000_000)
    prev = _prev_te_library(w, args.seed, max(1, int(0.5 * mbp))) if w is not None else []
    prev_len = np.array([len(q) for q in prev], dtype=np.float64)
    genome_ptr, contig_off = (w["genome"].data_ptr(), w["contig_off"]) if w is not None else (None, None)
    stage_ms = {}

    def lap(name, t0):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        stage_ms[name] = stage_ms.get(name, 0.0) + 1000.0 * (t1 - t0)
        return t1

    def inner_step(fresh=True):
        # fresh: every step computes the genome's minimizers -- nothing kept from the step before.  Inside the end-to-end step the build
        # follows the prev_TE step on the same packed chunk and redoes only the masked tiles (hite_copy_index_forget in include/hite_gpu.h)
        ctx.copy_index_build(fresh=fresh)
        (oc, os_, oe), st = ctx.coarse_stage_dev(1_000_000, sc, so, 4000, 30000)
        return st, (oc, os_, oe)

    def full_step():
        t0 = time.perf_counter()
        ctx.genome_pack_dev(genome_ptr, contig_off)
        t0 = lap("pack", t0)
        masked = ctx.tr_mask_dev(500)
        t0 = lap("tandem_mask", t0)
        n_masked_copies = 0
        if prev:
            # index of the tandem-masked chunk, for the library's look-ups only (as util._copies_as_blast6)
            cf, ct, s1, e1, _mn, _an = ctx.find_copies_table(prev, restricted=True)
            need = np.repeat(0.95 * prev_len, np.diff(cf))            # full-length copies: >= 95 % of the library sequence
            keep = (e1 - s1 + 1) >= need
            cc, ss, ee = ct[keep], s1[keep], e1[keep]
            ctx.genome_mask(cc, ss, ee)
            n_masked_copies = len(cc)
        t0 = lap("prev_te_mask", t0)
        st, (oc, os_, oe) = inner_step(fresh=False)    # (builds the index of the masked chunk first)
        t0 = lap("index_search_fmea", t0)
        nb = ctx.flanking_seq_dev(oc, os_, oe, 50)
        t0 = lap("flank_gather", t0)
        return st, len(oc), masked, n_masked_copies, nb

    # ---- inner number (as in rounds 2 and 3) --------------------------------------------------------------------------------
    for _ in range(2):
        inner_step()
    torch.cuda.synchronize()
    steps = max(1, min(3, args.steps))
    ctx.profile(on=True, reset=True)
    t1 = time.perf_counter()
    for _ in range(steps):
        st, iv = inner_step()
    torch.cuda.synchronize()
    ms_inner = 1000.0 * (time.perf_counter() - t1) / steps
    prof_inner = {k: round(v[0] / steps, 3) for k, v in sorted(ctx.profile(on=False).items())}
    alg = coarse_alg_bytes(st)
    inner = {"metric": "index + all-vs-all seeding + FMEA alone (the coarse number of rounds 2 and 3)", "ms_per_step": round(ms_inner, 3),
             "value": round(mbp / (ms_inner * 1e-3), 1), "unit": "Mbp/s", "seeds": st[0], "anchors": st[1], "clusters": st[2], "hsp_records": st[3],
             "repeat_intervals": len(iv[0]), "stages_ms": prof_inner,
             "roofline": {"bound": "hbm", "achieved": round(alg / (ms_inner * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                          "frac": round(alg / (ms_inner * 1e-3) / 1e9 / PEAK_HBM_GBS, 5),
                          "note": COARSE_ALG_NOTE}}
    if w is None:
        return dict(inner, inner=None)
    # ---- end to end -----------------------------------------------------------------------------------------------------
    # two untimed steps, as everywhere: the first grows the arenas of the index state to what the masked chunk needs, the second
    # consolidates them into one block
    for _ in range(2):
        full_step()
    stage_ms.clear()
    torch.cuda.synchronize()
    ctx.profile(on=True, reset=True)
    t1 = time.perf_counter()
    per_step = []
    for _ in range(steps):
        t2 = time.perf_counter()
        st, n_iv, masked, n_mc, nb = full_step()
        per_step.append(round(1000.0 * (time.perf_counter() - t2), 1))
    torch.cuda.synchronize()
    ms = 1000.0 * (time.perf_counter() - t1) / steps
    prof_full = {k: round(v[0] / steps, 3) for k, v in sorted(ctx.profile(on=False).items())}
    blk = {"metric": "coarse_boundary step END TO END (stage 3.1: pack + tandem masking + prev_TE masking + index + all-vs-all seeding + FMEA + "
                     "flanked sequences), whole genome as one chunk",
           "ms_per_step": round(ms, 3), "value": round(mbp / (ms * 1e-3), 1), "unit": "Mbp/s", "steps": steps, "per_step_ms": per_step, "library_stages_ms": prof_full,
           "stages_ms": {k: round(v / steps, 3) for k, v in stage_ms.items()},
           "tandem_masked_bases": int(masked), "prev_te_sequences": len(prev), "prev_te_copies_masked": int(n_mc),
           "hsp_records": st[3], "repeat_intervals": n_iv, "flanked_bytes": int(nb), "inner": inner}
    # leave the context as the headline left it: the unmasked genome and its index
    ctx.genome_pack_dev(genome_ptr, contig_off)
    ctx.copy_index_build()
    if with_cpu:
        import multiprocessing as mp
        threads = int(args.cpu_threads) if args.cpu_threads and args.cpu_threads > 0 else min(40, os.cpu_count() or 1)
        sub = min(mbp, 20)
        t0 = time.perf_counter()
        if threads == 1:
            res = [_coarse_cpu_worker((args.seed, sub))]
        else:
            with mp.get_context("spawn").Pool(threads) as pool:
                res = pool.map(_coarse_cpu_worker, [(args.seed + 31 * k, sub) for k in range(threads)])
        wall = time.perf_counter() - t0
        busy = max(r[0] for r in res)
        slow = max(res, key=lambda r: r[0])
        cpu = {"value": round(threads * sub / busy, 2), "unit": "Mbp/s", "cores": threads, "kind": "port",
               "stages_s_slowest_worker": {k: round(v, 2) for k, v in slow[3].items()},
               "sample": "%d workers x one %d Mbp sub-genome each, same family density (slowest worker %.1f s, %.1f s wall incl. start-up and generation): "
                         "%d HSP records -> %d intervals in all; CPU twins of the same stages in the same order (orc_tr_mask, orc_find_copies + "
                         "masking, orc_seed_allvsall + orc_fmea, flank slices)"
                         % (threads, sub, busy, wall, sum(r[1] for r in res), sum(r[2] for r in res))}
        blk["cpu_baseline"] = cpu
        blk["speedup_vs_cpu_baseline"] = round(blk["value"] / cpu["value"], 1) if cpu["value"] else None
    return blk


def c5_mode(args):
    """BASELINE.json configs[4]: panHiTE-style population genomes, ONE genome per GPU (replicas of the whole fine stage, no
    data-path collective), then the per-genome TE libraries are all-gathered (padded consensus pools + lengths, RCCL) and
    merged into one non-redundant library by deredundant_for_LTR_v5 (pan_remove_redundancy.py:16-46) on rank 0.
    A step = copy finding + fine stage on the rank's genome + the all-gather; the merge is timed separately (once)."""
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    backend, gpu_index, dev, cdev = rank_devices(torch, dist, rank, local_rank, world)
    import tempfile

    import hite_amd
    from hite_amd import dist as hd
    from hite_amd import synth, util
    from hite_amd._lib import CALL_DTYPE

    mbp, tir_d, ltr_d = CONFIGS["C5"]
    if args.genome_mbp is not None:
        mbp = args.genome_mbp
    n_tir = args.tir_families if args.tir_families is not None else int(tir_d * mbp)
    n_ltr = args.ltr_families if args.ltr_families is not None else int(ltr_d * mbp)
    # --genomes G on ONE process: the G population genomes one after another on this GPU (config C5 at its configured size without a
    # node: every genome the whole fine stage, then the real merge of G libraries); with several ranks: one genome per rank
    n_seq_genomes = max(1, args.genomes) if world == 1 else 1
    ctx = hite_amd.Context(gpu_index)
    stream = torch.cuda.Stream(device=dev)
    sp = stream.cuda_stream
    elapsed, total_cands, per_genome = 0.0, 0, []
    seqs_all, ranks_all = [], []
    for gi in range(n_seq_genomes):
        w = synth.make_workload(genome_bp=mbp * 1_000_000, n_tir=n_tir, n_ltr=n_ltr, cands_per_family=args.cands_per_family,
                                seed=args.seed + 1009 * (rank + gi + 1), device=dev, family_seed=args.seed, family_keep=0.7)
        n_cand = len(w["cand_off"]) - 1
        nbytes = int(w["cand_off"][-1])
        ctx.release_copy_index()
        ctx.genome_pack_dev(w["genome"].data_ptr(), w["contig_off"], sp)
        ctx.copy_index_build(sp)
        torch.cuda.synchronize()
        d_cand = torch.from_numpy(np.concatenate([w["cands"], np.zeros(64, np.uint8)])).to(dev)
        d_off = torch.from_numpy(np.ascontiguousarray(w["cand_off"])).to(dev)
        d_calls = torch.zeros(max(1, n_cand) * 32, dtype=torch.uint8, device=dev)
        cap = nbytes + 200 * n_cand + 4096
        d_cons = torch.zeros(cap + 64, dtype=torch.uint8, device=dev)

        def library():
            calls = d_calls.cpu().numpy().view(CALL_DTYPE)[:n_cand]
            pool = d_cons.cpu().numpy()
            return [pool[c["cons_off"]:c["cons_off"] + c["cons_len"]].tobytes() for c in calls if c["is_te"]]

        state = {"found": None, "n_copies": 0}

        def step(gather):
            nc, p_cf, p_ct, p_s1, p_e1, p_mn, _an = ctx.find_copies_dev(n_cand, d_cand.data_ptr(), d_off.data_ptr(), nbytes, sp)
            p_cl = ctx.copy_clips_dev()       # (records in the reference's coordinates: the rows are padded by the clipped candidate bases)
            state["found"], state["n_copies"] = (nc, p_cf, p_ct, p_s1, p_e1, p_mn, p_cl), nc
            ctx.flank_region_align_dev("tir", 1, n_cand, d_cand.data_ptr(), d_off.data_ptr(), p_cf, nc, p_ct, p_s1, p_e1, p_mn, 50,
                                       d_calls.data_ptr(), d_cons.data_ptr(), cap, sp, d_clip=p_cl)
            stream.synchronize()
            if not gather:
                return None
            mine = library()
            if world > 1:
                return hd.allgather_library(mine, device=cdev)
            return mine, np.zeros(len(mine), dtype=np.int64)

        for _ in range(2 + args.warmup):
            step(False)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        lib = None
        for _ in range(args.steps):
            lib = step(True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        e_g = time.perf_counter() - t1
        elapsed += e_g
        total_cands += n_cand
        per_genome.append({"genome": gi, "candidates": n_cand, "ms_per_step": round(1000.0 * e_g / max(1, args.steps), 3), "library_sequences": len(lib[0])})
        seqs_all += list(lib[0])
        ranks_all += [int(r) + gi for r in lib[1]]
        if gi + 1 < n_seq_genomes:       # (the last genome's workload stays: the CPU leg and the spot checks run on it)
            del w, d_cand, d_off, d_calls, d_cons
            torch.cuda.empty_cache()
    lib = (seqs_all, np.asarray(ranks_all, dtype=np.int64))
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tn = torch.tensor([total_cands], dtype=torch.int64, device=cdev)
        dist.all_reduce(tn)
        total_cands = int(tn.item())
    if rank == 0:
        seqs, ranks = lib
        tmp = tempfile.mkdtemp(prefix="hite_c5_")
        merged = os.path.join(tmp, "merged.fa")
        with open(merged, "w") as f:
            for i, (sq, r) in enumerate(zip(seqs, ranks)):
                f.write(">G%d-TE_%d#Unknown\n%s\n" % (int(r), i, sq.decode()))
        # host view of the step's candidates, copy table and calls for the CPU leg / the spot checks: taken NOW, the merge below
        # re-packs the context with the library and drops the copy index the table lives in
        calls = d_calls.cpu().numpy().view(CALL_DTYPE)[:n_cand].copy()
        cons_host = d_cons.cpu().numpy()
        wv = None
        if (world == 1 and not args.no_cpu_baseline) or args.verify > 0:
            wv = host_workload(w, ctx, state, 0, n_cand)
        t2 = time.perf_counter()
        util._CTX = ctx
        stages = {}
        out_path = util.deredundant_for_LTR_v5(merged, tmp, 1, "terminal", 0.95, 0, ctx=ctx, stages=stages)
        merge_s = time.perf_counter() - t2
        n_out = len(util.read_fasta(out_path)[0])
        n_final = len(util.read_fasta(merged + ".cons")[0])
        # ---- outside the timed region: the same-host CPU leg and the parity spot checks of THIS run --------------------------
        cpu, ver = None, None
        if world == 1 and not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline(wv, args.cpu_seconds, args.cpu_threads, True)
                if n_seq_genomes == 1:       # (the twins' all-vs-all of the library is super-linear: 39 s for one genome's library on one core)
                    cpu["merge"] = c5_merge_cpu(merged, tmp, util)
            except Exception as e:   # the GPU line must not depend on the CPU leg
                cpu = {"value": None, "unit": "candidates/s", "cores": args.cpu_threads, "kind": "port", "sample": "failed: %s: %s" % (type(e).__name__, e)}
        if args.verify > 0:
            ver = verify(wv, calls, cons_host, args.verify if world == 1 else min(args.verify, 64))
            ct = (cpu or {}).pop("copy_tables", None)
            if ct is not None:
                ver["copy_tables"] = ct
            try:
                ver["merge"] = c5_merge_verify(merged, tmp, util, ctx, stages, (cpu or {}).get("merge"))
            except Exception as e:
                ver["merge"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if cpu and "merge" in cpu:
            cpu["merge"].pop("_clusters", None)
            cpu["merge"].pop("_files", None)
        import shutil as _sh
        _sh.rmtree(tmp, ignore_errors=True)
        ms = 1000.0 * elapsed / max(1, args.steps)
        print(json.dumps({
            "metric": "candidate TE boundaries/sec, %d Mbp population genomes, %s (config C5), + merged non-redundant TE library" %
                      (mbp, "one genome per GPU" if n_seq_genomes == 1 else "%d genomes one after another on one GPU" % n_seq_genomes),
            "value": round(total_cands * args.steps / elapsed, 2), "unit": "candidates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": data_note(torch, backend),
            "config": {"workload": "C5: %d x %d Mbp genomes (%s), %d TIR + %d LTR families drawn from a shared pool (70 %% per genome), "
                                   "%d candidates in all judged as TIR; step = copy finding + fine stage + all-gather of the per-genome libraries"
                                   % (world * n_seq_genomes, mbp, "one per GPU" if n_seq_genomes == 1 else "one after another on this GPU: ms_per_step is the SUM over the genomes",
                                      n_tir, n_ltr, total_cands),
                       "genomes": world * n_seq_genomes, "per_genome": per_genome,
                       "library_sequences_in": len(seqs), "library_sequences_out": n_out, "library_sequences_final": n_final,
                       "library_clusters": len(stages.get("clusters", [])), "library_hits": stages.get("hits"), "merge_seconds": round(merge_s, 2),
                       "merge": "deredundant_for_LTR_v5 on rank 0 (library-vs-library seeding, chaining, clustering, star alignments, consensus)",
                       "parallelism": "replicas (one genome per GPU), all-gather of padded consensus pools"},
            "roofline": None, "cpu_baseline": cpu, "verify": ver}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def c5_merge_cpu(merged, tmp, util):
    """CPU leg of the library merge: the same host code (hite_amd/util.py deredundant_for_LTR_v5) with every device stage
    answered by its CPU twin (tests/oracle_ctx.py), on the WHOLE merged library, single thread, timed"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_ctx import OracleCtx

    twin = os.path.join(tmp, "twin_merged.fa")
    import shutil as _sh
    _sh.copyfile(merged, twin)
    st = {}
    t0 = time.perf_counter()
    util.deredundant_for_LTR_v5(twin, tmp, 1, "terminal", 0.95, 0, ctx=OracleCtx(), stages=st)
    dt = time.perf_counter() - t0
    return {"seconds": round(dt, 2), "cores": 1, "kind": "port", "clusters": len(st.get("clusters", [])),
            "sample": "the whole merged library through deredundant_for_LTR_v5 with the CPU twins of every device stage", "_clusters": st.get("clusters"),
            "_files": (open(twin + ".tmp.cons").read(), open(twin + ".cons").read())}


def c5_merge_verify(merged, tmp, util, ctx, stages, cpu_merge):
    """parity of the merge of THIS run: clusters and output files of the GPU run against the twin run (the CPU leg when it ran,
    else a twin run on a sub-library of 150 clusters)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_ctx import OracleCtx

    gpu_files = (open(merged + ".tmp.cons").read(), open(merged + ".cons").read())
    if cpu_merge and cpu_merge.get("_clusters") is not None:
        files = cpu_merge.get("_files")
        return {"scope": "whole library", "clusters_equal": cpu_merge["_clusters"] == stages.get("clusters"),
                "files_equal": files == gpu_files, "against": "the same host code with the CPU twins of every device stage (tests/oracle_ctx.py)"}
    names, seqs = util.read_fasta(merged)
    clusters = stages.get("clusters", [])
    rng = np.random.default_rng(11)
    keep = set(n for i in rng.permutation(len(clusters))[:150] for n in clusters[i])
    sub = {n: seqs[n] for n in names if n in keep}
    outs = []
    for tag, cx in (("gpu", ctx), ("cpu", OracleCtx())):
        path = os.path.join(tmp, "sub_%s.fa" % tag)
        util.store_fasta(sub, path)
        util.deredundant_for_LTR_v5(path, tmp, 1, "terminal", 0.95, 0, ctx=cx)
        outs.append((open(path + ".tmp.cons").read(), open(path + ".cons").read()))
    return {"scope": "sub-library: the %d members of 150 random clusters" % len(sub), "files_equal": outs[0] == outs[1],
            "against": "the same host code with the CPU twins of every device stage (tests/oracle_ctx.py)"}


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
    import torch.distributed as dist

    backend, gpu_index, dev, cdev = rank_devices(torch, dist, rank, local_rank, world)
    mbp, tir_d, ltr_d = CONFIGS[args.config]
    if args.genome_mbp is not None:
        mbp = args.genome_mbp
    G = mbp * 1_000_000
    n_tir = args.tir_families if args.tir_families is not None else max(1, int(tir_d * mbp))
    n_ltr = args.ltr_families if args.ltr_families is not None else max(0, int(ltr_d * mbp))
    # replicas: every rank searches its own genome (config 5 of BASELINE.json: one genome per GPU); no collective on the data path
    w = synth.make_workload(genome_bp=G, n_tir=n_tir, n_ltr=n_ltr, cands_per_family=1, seed=args.seed + 977 * rank, device=dev)
    ctx = hite_amd.Context(gpu_index)
    ctx.genome_pack_dev(w["genome"].data_ptr(), w["contig_off"])
    sc, so = ctx.seed_segments(1_000_000)

    def step():
        ctx.copy_index_build(fresh=True)            # the index is part of the step here (rebuilt on the same handle, from the genome: no kept minimizer tiles)
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
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        ms = 1000.0 * elapsed / max(1, args.steps)
        out = {"metric": "coarse_boundary step (stage 3.1: all-vs-all seeding + FMEA) on the %s synthetic genome" % ("1 Gbp" if mbp == 1000 else "%d Mbp" % mbp),
               "value": round(world * mbp * args.steps / elapsed, 2), "unit": "Mbp/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "u8", "data": data_note(torch, backend),
               "config": {"workload": "%s genome: %d Mbp, %d TIR + %d LTR families, 1 Mbp segments, one chunk%s" %
                                      (args.config, mbp, n_tir, n_ltr, "; one genome per GPU (replicas)" if world > 1 else ""),
                          "seeds": stats[0], "anchors": stats[1], "clusters": stats[2], "hsp_records": stats[3], "repeat_intervals": n_iv},
               "roofline": {"bound": "hbm", "achieved": round(coarse_alg_bytes(stats) / (elapsed / args.steps) / 1e9, 2),
                            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                            "note": COARSE_ALG_NOTE}}
        out["roofline"]["frac"] = round(out["roofline"]["achieved"] / PEAK_HBM_GBS, 5)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = coarse_cpu_baseline(args, min(mbp, 20))
        if world == 1 and not args.no_coarse:
            # stage 3.1 END TO END (pack + tandem masking + prev_TE masking + index + seeding + FMEA + flanked sequences): the block the
            # default line carries as "coarse", here without the fine stage around it (what tools/profile_coarse.sh profiles)
            out["end_to_end"] = coarse_block(ctx, args, mbp, n_tir, n_ltr, torch, False, w=w)
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
