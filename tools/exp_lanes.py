#!/usr/bin/env python
"""Experiment: K contexts ("lanes") on ONE GPU, each judging a length-balanced share of the same candidate batch from its own
host thread and stream -- do the kernels of the lanes fill each other's stalls?  Prints ms per whole batch for K = 1, 2, 3."""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import hite_amd
from hite_amd import dist as hd, synth
from hite_amd._lib import CALL_DTYPE

mbp = int(os.environ.get("MBP", 1000))
steps = int(os.environ.get("STEPS", 5))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
G = mbp * 1_000_000
seed = 20250927 + 3
w = synth.make_workload(genome_bp=G, n_tir=int(2.5 * mbp), n_ltr=int(2.5 * mbp), cands_per_family=10, seed=seed, device=dev, cand_seed=seed + 7919)
share_of = int(os.environ.get("SHARE_OF", 0))        # > 1: the batch is rank 0's strong-scaling share of this many ranks (bench.py --config C4share)
if share_of > 1:
    ids0, _ = hd.shard_candidates_balanced(w["cand_off"], w["copy_first"], 0, share_of)
    sub = {}
    sub["cands"], sub["cand_off"] = hd.gather_csr(w["cands"], w["cand_off"], ids0)
    cf64 = np.asarray(w["copy_first"], dtype=np.int64)
    for k_ in ("contig", "start1", "end1", "minus"):
        sub[k_], new_cf = hd.gather_csr(w[k_], cf64, ids0)
    sub["copy_first"] = new_cf.astype(np.int32)
    for k_, v_ in sub.items():
        w[k_] = v_
n_all = len(w["cand_off"]) - 1
print("candidates", n_all, flush=True)

def up(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

class Lane:
    def __init__(self, rank, world):
        self.ids, _ = hd.shard_candidates_balanced(w["cand_off"], w["copy_first"], rank, world)
        cands, cand_off = hd.gather_csr(w["cands"], w["cand_off"], self.ids)
        self.n = len(cand_off) - 1
        self.bytes = int(cand_off[-1])
        self.ctx = hite_amd.Context(0)
        self.stream = torch.cuda.Stream(device=dev)
        self.sp = self.stream.cuda_stream
        self.ctx.genome_pack_dev(w["genome"].data_ptr(), w["contig_off"], self.sp)
        torch.cuda.synchronize()
        self.ctx.copy_index_build(self.sp)
        torch.cuda.synchronize()
        self.d_cand = up(np.concatenate([cands, np.zeros(64, np.uint8)]))
        self.d_off = up(np.asarray(cand_off, dtype=np.int64))
        self.d_calls = torch.zeros(self.n * 32, dtype=torch.uint8, device=dev)
        self.cap = self.bytes + 200 * self.n + 4096
        self.d_cons = torch.zeros(self.cap + 64, dtype=torch.uint8, device=dev)
    def step(self):
        c = self.ctx
        nc, p_cf, p_ct, p_s1, p_e1, p_mn, _ = c.find_copies_dev(self.n, self.d_cand.data_ptr(), self.d_off.data_ptr(), self.bytes, self.sp)
        c.flank_region_align_dev("tir", 1, self.n, self.d_cand.data_ptr(), self.d_off.data_ptr(), p_cf, nc, p_ct, p_s1, p_e1, p_mn, 50,
                                 self.d_calls.data_ptr(), self.d_cons.data_ptr(), self.cap, self.sp, d_clip=c.copy_clips_dev())
        self.stream.synchronize()

ref = None
for K in [int(x) for x in os.environ.get("LANES", "1,2,3").split(",")]:
    lanes = [Lane(r, K) for r in range(K)]
    def run_all():
        if K == 1:
            lanes[0].step(); return
        th = [threading.Thread(target=l.step) for l in lanes]
        for t in th: t.start()
        for t in th: t.join()
    for _ in range(4):
        run_all()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run_all()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    calls = np.zeros(n_all, dtype=CALL_DTYPE)
    for l in lanes:
        calls[l.ids] = l.d_calls.cpu().numpy().view(CALL_DTYPE)[: l.n]
    same = None
    if ref is None:
        ref = calls
    else:
        same = bool((calls.tobytes() == ref.tobytes()))
    print("lanes %d: %.2f ms per batch, %.0f candidates/s, calls identical to 1 lane: %s" % (K, ms, n_all / ms * 1e3, same), flush=True)
    del lanes
    torch.cuda.empty_cache()
