"""Multi-GPU plumbing for the fine stage: candidates are independent units, so they are
sharded across ranks with the packed genome replicated; the only exchange is ONE all-gather of the
fixed-size 32-byte call records (+ one of the packed consensus bytes when the caller wants them).
One process per GPU, torch.distributed ("nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
The reference has no counterpart: it fans candidates out over a fork()ed process pool
(/root/reference/module/Util.py:8141-8147)."""
import numpy as np

try:        # the single-GPU stages (coarse_boundary.py without torchrun) need numpy and the library only
    import torch
    import torch.distributed as dist
except ImportError:        # pragma: no cover
    torch = None
    dist = None

from ._lib import CALL_DTYPE


def process_group_state(group=None):
    """(a process group is up, rank, world); (False, 0, 1) without torch or without an initialised group"""
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return False, 0, 1
    return True, dist.get_rank(group), dist.get_world_size(group)


def collective_device(group=None, ctx=None):
    """where a collective's tensors have to live: RCCL ("nccl") moves device memory only -- the GPU of this rank's context --
    gloo moves host memory"""
    if dist is not None and dist.is_initialized() and dist.get_backend(group) == "nccl":
        dev = getattr(ctx, "device", None)
        return torch.device("cuda", int(dev) if dev is not None else torch.cuda.current_device())
    return "cpu"


def shard_bounds(n_items, world):
    """contiguous block partition: rank r owns [b[r], b[r+1])"""
    base, rem = divmod(int(n_items), int(world))
    sizes = [base + (1 if r < rem else 0) for r in range(world)]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def shard_candidates(cand_off, copy_first, rank, world):
    """slice the candidate CSR + copy CSR for one rank -> (c0, c1, cand byte range, copy range)"""
    b = shard_bounds(len(cand_off) - 1, world)
    c0, c1 = int(b[rank]), int(b[rank + 1])
    return c0, c1, (int(cand_off[c0]), int(cand_off[c1])), (int(copy_first[c0]), int(copy_first[c1]))


# ---- length-balanced block-cyclic sharding -----------------------------------------------------------------
# The step time of a rank is set by its longest alignments (the kernels are chains of the longest wavefront), so a contiguous
# block split lets one rank hold the long candidates.  Candidates are ordered by cost (bases x copies, descending; ties by
# index) and dealt to the ranks in a boustrophedon: position p of that order goes to rank p % world on even rounds and to
# world - 1 - p % world on odd ones.  Every rank computes the same assignment from the replicated CSR arrays, so the merge
# needs no exchange of indices: the all-gathered records are put back in candidate order by the inverse permutation.
def balanced_assignment(cost, world):
    """-> list of int64 arrays: the candidate ids of each rank, ascending"""
    cost = np.asarray(cost, dtype=np.int64)
    n = len(cost)
    order = np.lexsort((np.arange(n), -cost))          # cost descending, index ascending
    pos = np.arange(n)
    rnd, k = pos // world, pos % world
    rank_of_pos = np.where(rnd % 2 == 0, k, world - 1 - k)
    return [np.sort(order[rank_of_pos == r]).astype(np.int64) for r in range(world)]


def candidate_cost(cand_off, copy_first):
    """bases x (copies + 1) of every candidate: what its alignments cost, to first order"""
    cand_off = np.asarray(cand_off, dtype=np.int64)
    copy_first = np.asarray(copy_first, dtype=np.int64)
    return np.diff(cand_off) * (np.diff(copy_first) + 1)


def gather_csr(buf, off, ids):
    """rows `ids` of a CSR (buf, off) -> (buf', off') in that order"""
    off = np.asarray(off, dtype=np.int64)
    ids = np.asarray(ids, dtype=np.int64)
    lens = off[ids + 1] - off[ids]
    new_off = np.zeros(len(ids) + 1, dtype=np.int64)
    np.cumsum(lens, out=new_off[1:])
    if len(ids) == 0 or new_off[-1] == 0:
        return np.asarray(buf)[:0].copy(), new_off
    src = np.repeat(off[ids] - new_off[:-1], lens) + np.arange(new_off[-1], dtype=np.int64)
    return np.asarray(buf)[src], new_off


def shard_candidates_balanced(cand_off, copy_first, rank, world):
    """-> (ids of this rank, ids of every rank): the length-balanced block-cyclic share of the candidate batch"""
    shares = balanced_assignment(candidate_cost(cand_off, copy_first), world)
    return shares[rank], shares


# Buffers of the per-step merge, kept between steps (the send pad, the gathered records, the merged records, the inverse
# permutation): a step that allocates and concatenates them anew pays the allocator and a device synchronisation inside the timed
# region.  Keyed by (purpose, device); grow-only.  The tensors returned by the merges below alias them: valid until the next merge.
_MERGE_BUF = {}


def _merge_buf(key, n, dtype, device):
    k = (key, str(device), dtype)
    t = _MERGE_BUF.get(k)
    if t is None or t.numel() < n:
        t = torch.empty(max(n, 1), dtype=dtype, device=device)
        _MERGE_BUF[k] = t
    return t[:n]


def allgather_calls_balanced(local_calls, shares, group=None, alias=False):
    """local_calls: uint8 tensor (len(shares[rank]) * 32) in the order of shares[rank].  ONE padded all_gather_into_tensor, then
    the inverse permutation: returns the records of all candidates in candidate order -- a tensor of the caller's own; alias=True
    (the bench's timed loop): a view of a buffer kept between steps, which the NEXT merge overwrites."""
    world = dist.get_world_size(group)
    dev = local_calls.device
    max_n = max(1, max(len(s) for s in shares))
    pad = _merge_buf("pad", max_n * 32, torch.uint8, dev)
    pad[: local_calls.numel()] = local_calls            # (the tail of the pad is never read: every rank's count is known)
    out = _merge_buf("gathered", world * max_n * 32, torch.uint8, dev)
    dist.all_gather_into_tensor(out, pad, group=group)
    n_total = sum(len(s) for s in shares)
    cached = _MERGE_BUF.get("perm")      # ONE slot: the permutation of the shares seen last (a run's shares do not change between its steps)
    if cached is None or cached[0] is not shares or cached[1].device != dev:
        # row of candidate c in the gathered buffer
        src = np.empty(n_total, dtype=np.int64)
        for r, ids in enumerate(shares):
            src[ids] = r * max_n + np.arange(len(ids))
        cached = (shares, torch.from_numpy(src).to(dev))
        _MERGE_BUF["perm"] = cached
    merged = _merge_buf("merged", n_total * 32, torch.uint8, dev).view(n_total, 32)
    torch.index_select(out.view(world * max_n, 32), 0, cached[1], out=merged)
    return merged.reshape(-1) if alias else merged.reshape(-1).clone()


def allgather_consensus_balanced(local_calls_np, local_cons, shares, group=None):
    """as allgather_consensus for balanced shares: -> (calls in candidate order with cons_off rebased into the merged pool, pool)"""
    world = dist.get_world_size(group)
    dev = local_cons.device
    used = int(local_calls_np["cons_len"][local_calls_np["is_te"] != 0].sum())
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, torch.tensor([used], dtype=torch.int64, device=dev), group=group)
    sizes = sizes.cpu().numpy()
    mx = max(int(sizes.max()), 1)
    pad = torch.zeros(mx, dtype=torch.uint8, device=dev)
    pad[:used] = local_cons[:used]
    out = torch.empty(world * mx, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, pad, group=group)
    calls_t = torch.from_numpy(local_calls_np.view(np.uint8).copy()).to(dev)
    allc = allgather_calls_balanced(calls_t, shares, group).cpu().numpy().view(CALL_DTYPE).copy()
    base = np.concatenate([[0], np.cumsum(sizes)])
    for r, ids in enumerate(shares):
        allc["cons_off"][ids] += base[r]
    cons = torch.cat([out[r * mx: r * mx + int(sizes[r])] for r in range(world)])
    return allc, cons


def allgather_calls(local_calls, n_total, group=None, alias=False):
    """local_calls: uint8 tensor (n_local * 32) on the backend's device.  Returns the n_total records of all
    ranks in candidate order (block partition => rank order) -- a tensor of the caller's own; alias=True: a view of a buffer kept
    between steps, which the next merge overwrites.  One padded all_gather_into_tensor."""
    world = dist.get_world_size(group)
    dev = local_calls.device
    b = shard_bounds(n_total, world)
    max_n = max(1, int(np.max(np.diff(b))))
    pad = _merge_buf("pad", max_n * 32, torch.uint8, dev)
    pad[: local_calls.numel()] = local_calls
    out = _merge_buf("gathered", world * max_n * 32, torch.uint8, dev)
    dist.all_gather_into_tensor(out, pad, group=group)
    merged = _merge_buf("merged", n_total * 32, torch.uint8, dev)
    for r in range(world):                  # block partition: rank order is candidate order; no allocation, no concatenation
        n_r = int(b[r + 1] - b[r]) * 32
        merged[int(b[r]) * 32: int(b[r]) * 32 + n_r] = out[r * max_n * 32: r * max_n * 32 + n_r]
    return merged if alias else merged.clone()


def allgather_consensus(local_calls_np, local_cons, n_total, group=None):
    """gather the packed consensus pools; returns (calls (numpy CALL_DTYPE, cons_off rebased), cons bytes)"""
    world = dist.get_world_size(group)
    dev = local_cons.device
    used = int(local_calls_np["cons_len"][local_calls_np["is_te"] != 0].sum())
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([used], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, mine, group=group)
    sizes = sizes.cpu().numpy()
    mx = int(sizes.max())
    pad = torch.zeros(max(mx, 1), dtype=torch.uint8, device=dev)
    pad[:used] = local_cons[:used]
    out = torch.empty(world * max(mx, 1), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, pad, group=group)
    calls_t = torch.from_numpy(local_calls_np.view(np.uint8).copy()).to(dev)
    allc = allgather_calls(calls_t, n_total, group).cpu().numpy().view(CALL_DTYPE).copy()
    b = shard_bounds(n_total, world)
    base = np.concatenate([[0], np.cumsum(sizes)])
    cons = torch.cat([out[r * max(mx, 1): r * max(mx, 1) + int(sizes[r])] for r in range(world)])
    for r in range(world):
        allc["cons_off"][b[r]:b[r + 1]] += base[r]
    return allc, cons


# ---- config C5 (panHiTE: one genome per GPU): the per-rank TE libraries become one library on every rank ----------------------
def allgather_varlen(local, group=None):
    """local: 1-D tensor (any length, same dtype on every rank) -> (concatenation in rank order, per-rank lengths).  Two
    collectives: the lengths, then ONE padded all_gather_into_tensor."""
    world = dist.get_world_size(group)
    dev = local.device
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, torch.tensor([local.numel()], dtype=torch.int64, device=dev), group=group)
    sizes = sizes.cpu().numpy()
    mx = max(int(sizes.max()), 1)
    pad = torch.zeros(mx, dtype=local.dtype, device=dev)
    pad[: local.numel()] = local
    out = torch.empty(world * mx, dtype=local.dtype, device=dev)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * mx: r * mx + int(sizes[r])] for r in range(world)]), sizes


def allgather_library(seqs, device="cpu", group=None):
    """seqs: this rank's library (list of bytes / str) -> the libraries of all ranks as (list of bytes, rank of each), in rank
    order: the all-gather of padded consensus pools + their lengths that SURVEY 8e describes for config C5"""
    sb = [x.encode() if isinstance(x, str) else bytes(x) for x in seqs]
    pool = torch.from_numpy(np.frombuffer(b"".join(sb) + b"", dtype=np.uint8).copy()).to(device)
    lens = torch.tensor([len(x) for x in sb], dtype=torch.int64, device=device)
    all_pool, _ = allgather_varlen(pool, group)
    all_lens, counts = allgather_varlen(lens, group)
    all_pool = all_pool.cpu().numpy()
    all_lens = all_lens.cpu().numpy()
    off = np.concatenate([[0], np.cumsum(all_lens)])
    out = [all_pool[off[i]:off[i + 1]].tobytes() for i in range(len(all_lens))]
    ranks = np.repeat(np.arange(len(counts)), counts)
    return out, ranks


# ---- stage 3.1 (coarse_boundary) sharded over the ranks of a node ------------------------------------------
# The reference's units are the query FILES of the all-vs-all search: every get_longest_repeats_v4 call has its own first-come
# de-duplication and the results are unioned by name in file order (Util.py:4155, 4787-4794).  With the packed genome and its
# minimizer index replicated on every GPU:
#   1. seeding: rank r computes the HSPs of ITS share of the anchors -- a range of (strand, diagonal) whose edges never cut a
#      cluster (hite_seed_shard) -- so the union over the ranks is the unsharded table, record for record;
#   2. ONE all-to-all of 48-byte HSP records routes every record to the owner of its query file (files dealt round-robin);
#      the owner puts the pieces in rank order and sorts them stably by (query segment, subject segment): its slice of the
#      unsharded table, in the unsharded order (FMEA is order dependent);
#   3. FMEA per owned query file (hite_fmea_chain);
#   4. ONE all-gather of the interval lists (file id, contig, start, end), merged in file order, first name wins -- the
#      reference's dict union.
# The result equals the single-rank result by construction; tests/test_dist_gloo.py checks it on two gloo ranks with the CPU twins.
def query_files_of_segments(seg_len_each, base_threshold=1_000_000):
    """split_and_store_sequences (Util.py:4987) on the segment lengths: consecutive segments are collected until their total
    reaches base_threshold -> file id of every segment"""
    out = np.zeros(len(seg_len_each), dtype=np.int64)
    f, count = 0, 0
    for i, L in enumerate(seg_len_each):
        out[i] = f
        count += int(L)
        if count >= base_threshold:
            f += 1
            count = 0
    return out


def _exchange_rows(rows, dest, group=None, device="cpu"):
    """rows: int64 [n, k]; dest: rank of every row -> the rows sent to this rank by all ranks, in source-rank order (the order
    inside a source is kept).  all_to_all_single where the backend has it (RCCL), else a padded all-gather (gloo)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    k = rows.shape[1] if rows.ndim == 2 else 1
    order = np.argsort(dest, kind="stable")
    send = np.ascontiguousarray(rows[order])
    counts = np.bincount(dest, minlength=world).astype(np.int64)
    t_counts = torch.from_numpy(counts).to(device)
    all_counts = torch.empty(world * world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(all_counts, t_counts, group=group)
    all_counts = all_counts.cpu().numpy().reshape(world, world)          # [source, destination]
    recv_counts = all_counts[:, rank]
    t_send = torch.from_numpy(send.reshape(-1)).to(device)
    if dist.get_backend(group) == "nccl":
        t_recv = torch.empty(int(recv_counts.sum()) * k, dtype=torch.int64, device=device)
        dist.all_to_all_single(t_recv, t_send, output_split_sizes=[int(c) * k for c in recv_counts],
                               input_split_sizes=[int(c) * k for c in counts], group=group)
        return t_recv.cpu().numpy().reshape(-1, k)
    gathered, sizes = allgather_varlen(t_send, group)
    gathered = gathered.cpu().numpy().reshape(-1, k)
    src_first = np.concatenate([[0], np.cumsum(all_counts.sum(axis=1))])
    parts = []
    for src in range(world):
        a = src_first[src] + int(all_counts[src, :rank].sum())
        parts.append(gathered[a:a + int(all_counts[src, rank])])
    return np.concatenate(parts) if parts else gathered[:0]


def coarse_stage_sharded(ctx, seg_len, skip_gap, max_len, group=None, device=None, base_threshold=1_000_000, seg_table=None):
    """stage 3.1 on the genome resident in `ctx` (the same on every rank), sharded as described above.
    -> (contig ids, starts, ends) of the repeat intervals in the single-rank order, identical on every rank.
    Without an initialised process group: the single-rank computation (the same code path, no collectives).
    seg_table = (chromosome id, offset) of every packed sequence when those are 'chr$offset' segments of a chunk file
    (determine_repeat_boundary_v5: the intervals then come out in chromosome coordinates); default: the packed contigs cut every seg_len.
    device = where the collectives' tensors live; None: the GPU of `ctx` under RCCL, the host under gloo."""
    multi, rank, world = process_group_state(group)
    if device is None:          # the backend decides: device tensors for RCCL, host tensors for gloo
        device = collective_device(group, ctx)
    clen = np.asarray(ctx.contig_len, dtype=np.int64)
    if seg_table is None:
        seg_chrom, seg_off = ctx.seed_segments(seg_len)
        seg_lens = np.minimum(seg_len, clen[seg_chrom] - seg_off)
    else:
        seg_chrom, seg_off = np.asarray(seg_table[0], dtype=np.int32), np.asarray(seg_table[1], dtype=np.int64)
        seg_lens = clen                 # one segment per packed sequence
    file_of = query_files_of_segments(seg_lens, base_threshold)
    n_files = int(file_of[-1]) + 1 if len(file_of) else 0
    ctx.seed_shard(rank, world)
    try:
        tab = ctx.seed_allvsall(seg_len=seg_len)
    finally:
        ctx.seed_shard(0, 0)
    rows = np.stack([np.asarray(tab[k], dtype=np.int64) for k in ("qseg", "sseg", "qs", "qe", "ss", "se")], axis=1) if len(tab["qseg"]) \
        else np.zeros((0, 6), dtype=np.int64)
    if multi and world > 1:
        rows = _exchange_rows(rows, (file_of[rows[:, 0]] % world).astype(np.int64), group, device)
    order = np.argsort(rows[:, 0] * (len(seg_chrom) + 1) + rows[:, 1], kind="stable")
    rows = rows[order]
    fkey = file_of[rows[:, 0]] if len(rows) else np.zeros(0, dtype=np.int64)
    out_f, out_c, out_s, out_e = [], [], [], []
    for f in range(rank, n_files, world):
        a, b = np.searchsorted(fkey, f, "left"), np.searchsorted(fkey, f, "right")
        if b <= a:
            continue
        r = rows[a:b]
        oc, os_, oe = ctx.fmea_chain(r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4], r[:, 5], seg_chrom, seg_off, skip_gap, max_len)
        out_f.append(np.full(len(oc), f, dtype=np.int64)); out_c.append(np.asarray(oc, dtype=np.int64))
        out_s.append(np.asarray(os_, dtype=np.int64)); out_e.append(np.asarray(oe, dtype=np.int64))
    mine = np.stack([np.concatenate(x) if x else np.zeros(0, dtype=np.int64) for x in (out_f, out_c, out_s, out_e)], axis=1)
    if multi and world > 1:
        allv, _ = allgather_varlen(torch.from_numpy(np.ascontiguousarray(mine).reshape(-1)).to(device), group)
        mine = allv.cpu().numpy().reshape(-1, 4)
    mine = mine[np.argsort(mine[:, 0], kind="stable")]          # file order; inside a file the owner's order
    seen, keep = set(), []
    for i, (_f, c, s_, e_) in enumerate(mine.tolist()):
        if (c, s_, e_) not in seen:
            seen.add((c, s_, e_))
            keep.append(i)
    mine = mine[keep]
    return mine[:, 1].astype(np.int32), mine[:, 2].copy(), mine[:, 3].copy()
