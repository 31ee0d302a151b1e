"""helper of tests/test_gpu_parity.py::test_coarse_stage_sharded_two_ranks_on_one_gpu: started twice by torch.distributed.run; every
rank owns a context on GPU 0 with the same packed genome, the ranks exchange HSP records and interval lists over gloo; rank 0
writes the merged result as JSON to argv[1]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch.distributed as dist  # noqa: E402

import hite_amd  # noqa: E402
import synth_small  # noqa: E402
from hite_amd import dist as hd  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
ctx = hite_amd.Context(0)
try:
    g = synth_small.make(31, n_fam=14, n_chr=3, chr_len=150_000)
    ctx.genome_pack(g["contigs"])
    shard = hd.coarse_stage_sharded(ctx, 50_000, 2000, 30000, base_threshold=100_000)
    if rank == 0:
        json.dump([np.asarray(x).tolist() for x in shard], open(sys.argv[1], "w"))
finally:
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()
