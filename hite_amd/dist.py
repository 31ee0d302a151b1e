"""Multi-GPU plumbing for the fine stage: candidates are independent units, so they are
sharded across ranks with the packed genome replicated; the only exchange is ONE all-gather of the
fixed-size 32-byte call records (+ one of the packed consensus bytes when the caller wants them).
One process per GPU, torch.distributed ("nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
The reference has no counterpart: it fans candidates out over a fork()ed process pool
(/root/reference/module/Util.py:8141-8147)."""
import numpy as np
import torch
import torch.distributed as dist

from ._lib import CALL_DTYPE


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


def allgather_calls(local_calls, n_total, group=None):
    """local_calls: uint8 tensor (n_local * 32) on the backend's device.  Returns the n_total records of all
    ranks in candidate order (block partition => rank order).  One padded all_gather_into_tensor."""
    world = dist.get_world_size(group)
    b = shard_bounds(n_total, world)
    max_n = int(np.max(np.diff(b)))
    pad = torch.zeros(max_n * 32, dtype=torch.uint8, device=local_calls.device)
    pad[: local_calls.numel()] = local_calls
    out = torch.empty(world * max_n * 32, dtype=torch.uint8, device=local_calls.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = [out[r * max_n * 32: r * max_n * 32 + int(b[r + 1] - b[r]) * 32] for r in range(world)]
    return torch.cat(parts)


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
