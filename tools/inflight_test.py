"""Experiment: two half-batches in flight on one GPU (two contexts, two host threads) against one full batch.
usage: python tools/inflight_test.py [steps]"""
import os, sys, time, threading
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hite_amd
from hite_amd import dist as hd, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
G = 1000 * 1_000_000
w = synth.make_workload(genome_bp=G, n_tir=2500, n_ltr=2500, cands_per_family=10, seed=20250927 + 3, device=dev, cand_seed=20250927 + 3 + 7919)

def up(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

lens_all = np.diff(w["cand_off"])
order_len = np.argsort(-lens_all, kind="stable")
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0     # > 0: part 0 = the longest `frac` of the candidates, part 1 = the rest

class Half:
    def __init__(self, rank, world):
        if frac > 0 and world == 2:
            k = int(len(order_len) * frac)
            sel = np.sort(order_len[:k] if rank == 0 else order_len[k:])
            offs = w["cand_off"]
            pieces = [w["cands"][offs[c]:offs[c + 1]] for c in sel]
            self.n = len(sel)
            cand_bytes = np.concatenate(pieces)
            cand_off = np.zeros(self.n + 1, np.int64); np.cumsum([len(x) for x in pieces], out=cand_off[1:])
            self.bytes = int(cand_off[-1])
            self._init_rest(cand_bytes, cand_off)
            return
        c0, c1, (b0, b1), (k0, k1) = hd.shard_candidates(w["cand_off"], w["copy_first"], rank, world)
        self.n = c1 - c0
        self.bytes = b1 - b0
        self._init_rest(w["cands"][b0:b1], w["cand_off"][c0:c1 + 1] - b0)

    def _init_rest(self, cand_bytes, cand_off):
        self.ctx = hite_amd.Context(0)
        self.stream = torch.cuda.Stream(device=dev)
        self.sp = self.stream.cuda_stream
        self.ctx.genome_pack_dev(w["genome"].data_ptr(), w["contig_off"], self.sp)
        torch.cuda.synchronize()
        self.ctx.copy_index_build(self.sp)
        torch.cuda.synchronize()
        self.d_calls = torch.zeros(max(1, self.n) * 32, dtype=torch.uint8, device=dev)
        self.cons_cap = self.bytes + 200 * self.n + 4096
        self.d_cons = torch.zeros(self.cons_cap + 64, dtype=torch.uint8, device=dev)
        self.d_cand = up(np.concatenate([cand_bytes, np.zeros(64, np.uint8)]))
        self.d_cand_off = up(cand_off)
    def step(self):
        nc, p_cf, p_ct, p_s1, p_e1, p_mn, _ = self.ctx.find_copies_dev(self.n, self.d_cand.data_ptr(), self.d_cand_off.data_ptr(), self.bytes, self.sp)
        self.ctx.flank_region_align_dev("tir", 1, self.n, self.d_cand.data_ptr(), self.d_cand_off.data_ptr(), p_cf, nc, p_ct, p_s1, p_e1, p_mn,
                                        50, self.d_calls.data_ptr(), self.d_cons.data_ptr(), self.cons_cap, self.sp, d_clip=self.ctx.copy_clips_dev())
        self.stream.synchronize()

halves = [Half(r, parts) for r in range(parts)]
for h in halves:
    for _ in range(3):
        h.step()
torch.cuda.synchronize()

def run(h, k):
    for _ in range(k):
        h.step()

# sequential
t0 = time.perf_counter()
for _ in range(steps):
    for h in halves:
        h.step()
torch.cuda.synchronize()
seq = (time.perf_counter() - t0) / steps
# concurrent
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(h, steps)) for h in halves]
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize()
con = (time.perf_counter() - t0) / steps
from hite_amd._lib import CALL_DTYPE
te = sum(int((h.d_calls.cpu().numpy().view(CALL_DTYPE)[:h.n]["is_te"] != 0).sum()) for h in halves)
print("parts %d: sequential %.2f ms per full batch, concurrent %.2f ms per full batch, TE calls %d" % (parts, 1000 * seq, 1000 * con, te))
