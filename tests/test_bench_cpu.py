"""CPU-side checks of bench.py: the cpu_baseline leg (the oracle chain is test / measurement infrastructure, the only
part of bench.py that may touch it) on a tiny workload, scalar and multi-process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_cpu_baseline_leg_runs_on_a_tiny_workload():
    import torch

    import bench
    from hite_amd import synth

    w = synth.make_workload(genome_bp=2_000_000, n_tir=6, n_ltr=4, cands_per_family=2, seed=11, device=torch.device("cpu"))
    n_cand = len(w["cand_off"]) - 1
    wv = bench.host_workload(w, None, None, 0, n_cand)       # the generator's copy table
    for threads in (1, 2):
        r = bench.cpu_baseline(wv, 1.0, threads)
        assert set(r) == {"value", "unit", "cores", "kind", "sample", "reference_python_vs_c_port"}
        assert r["cores"] == threads and r["kind"] == "port" and r["unit"] == "candidates/s" and r["value"] > 0
    r = bench.cpu_baseline(wv, 0.5, 1, with_copies=True)     # + the CPU twin of the copy finder, charged per candidate
    assert r["copy_finding"]["index_s"] > 0 and r["value"] > 0
    # the twin's copy table is kept and compared with the table of the step (here: the generator's, which differs -- the
    # comparison itself is what is exercised; on the GPU box the step's table comes from the HIP finder and must agree)
    assert r["copy_tables"]["checked"] > 0 and r["copy_tables"]["copies_in_sample"] > 0


def test_coarse_cpu_worker_runs():
    """the CPU leg of the coarse block of the default bench line: the twins of stage 3.1 on one small sub-genome"""
    import bench

    dt, n_hsp, n_iv, stages, n_masked = bench._coarse_cpu_worker((5, 2))
    assert dt > 0 and n_hsp > 0 and n_iv > 0
    # stage 3.1 end to end: tandem masking, prev_TE masking, search + FMEA, flanked sequences -- every leg ran
    assert set(stages) == {"tandem", "prev_te", "search_fmea", "flank"} and all(v > 0 for v in stages.values()) and n_masked >= 0


def test_bench_shards_like_the_library():
    """--scaling strong uses hite_amd.dist.shard_candidates: the shares tile the batch"""
    import numpy as np

    from hite_amd import dist as hd

    cand_off = np.arange(0, 1010, 10)
    copy_first = np.arange(0, 303, 3)
    seen = []
    for r in range(3):
        c0, c1, (b0, b1), (k0, k1) = hd.shard_candidates(cand_off, copy_first, r, 3)
        assert (b0, b1) == (10 * c0, 10 * c1) and (k0, k1) == (3 * c0, 3 * c1)
        seen += list(range(c0, c1))
    assert seen == list(range(100))


def test_bench_refuses_to_run_without_a_gpu():
    """no CPU fallback: without a GPU the bench exits with a message instead of timing something else"""
    import subprocess

    import torch

    if torch.cuda.is_available():
        return
    rc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                        capture_output=True, text=True)
    assert rc.returncode != 0 and "needs a GPU" in (rc.stdout + rc.stderr)
