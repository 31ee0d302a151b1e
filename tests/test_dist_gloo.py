"""CPU, world_size 2, gloo: the multi-GPU merge path (shard -> judge locally -> all-gather of 32-byte call
records and consensus pools) gives exactly the single-process result.  The per-rank "judge" here is a
deterministic stand-in that fabricates records from candidate ids (no GPU in this container); the GPU
kernels themselves are covered by the -m gpu tests."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _fake_calls(c0, c1):
    from hite_amd._lib import CALL_DTYPE

    n = c1 - c0
    calls = np.zeros(n, dtype=CALL_DTYPE)
    ids = np.arange(c0, c1)
    calls["is_te"] = (ids % 3 != 0)
    calls["info"] = ids % 4
    calls["row_num"] = ids % 101
    calls["bstart"] = 50 + ids % 7
    calls["bend"] = 900 + ids % 11
    calls["cons_len"] = np.where(calls["is_te"] != 0, 5 + ids % 9, 0)
    off = np.concatenate([[0], np.cumsum(calls["cons_len"])])
    calls["cons_off"] = off[:-1]
    cons = np.zeros(int(off[-1]), dtype=np.uint8)
    for i, cid in enumerate(ids):
        L = int(calls["cons_len"][i])
        cons[off[i]:off[i] + L] = (cid * 7 + np.arange(L)) % 251
    return calls, cons


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hite_amd import dist as hd

    b = hd.shard_bounds(n_total, world)
    calls, cons = _fake_calls(int(b[rank]), int(b[rank + 1]))
    allc, allcons = hd.allgather_consensus(calls, torch.from_numpy(cons), n_total)
    if rank == 0:
        q.put((allc.tobytes(), allcons.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_calls_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n_total = 1001  # ragged split
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got_calls, got_cons = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from hite_amd._lib import CALL_DTYPE

    exp_calls, exp_cons = _fake_calls(0, n_total)
    got = np.frombuffer(got_calls, dtype=CALL_DTYPE)
    assert np.array_equal(got, exp_calls)
    assert got_cons == exp_cons.tobytes()


def test_shard_bounds():
    from hite_amd import dist as hd

    for n in (0, 1, 7, 8, 1001):
        for w in (1, 2, 3, 8):
            b = hd.shard_bounds(n, w)
            assert b[0] == 0 and b[-1] == n and len(b) == w + 1
            assert (np.diff(b) >= 0).all() and np.diff(b).max() - np.diff(b).min() <= 1
