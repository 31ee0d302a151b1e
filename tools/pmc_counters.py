#!/usr/bin/env python
"""Per-kernel SQ counters of one rocprofv3 PMC pass -> profiles/r02_sq_counters.json (+ a text table).
usage: pmc_counters.py <rocprof_out_dir> <bench_json_of_the_same_run> <out_json> <out_txt> [header]
The bench line of the profiled run gives the steps of the run (timed + untimed) and the pair-columns per step, so that the
json can state wave-level vector instructions per step and per pair-column for every alignment kernel; bench.py reads it."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import per_kernel  # noqa: E402

COUNTERS = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_INSTS_VMEM"]
STAGE = {"align_fwd4_kernel": "align_fwd4", "align_fwd_kernel<4>": "align_fwd4", "align_fwd_kernel<8>": "align_fwd_wide8", "align_fwd_kernel<16>": "align_fwd_wide16",
         "align_fwd_kernel<32>": "align_fwd_wide32", "align_fwd8_pair_kernel": "align_fwd_wide8", "align_tb_kernel": "align_tb",
         "align_fwd_lanes_kernel<4>": "align_fwd4_lanes", "align_fwd_lanes_kernel<8>": "align_fwd_wide_lanes", "align_tb_strips_kernel": "align_tb_strips",
         "align_wide_fwd_kernel": "align_fallback_fwd", "align_wide_tb_kernel": "align_fallback_tb", "align_planes_kernel": "align_prep_planes"}
FWD4_LAUNCHES_PER_STEP = 2      # one pipeline step = the first500 + last500 pass and the full-length pass: align_fwd4_kernel runs once in each


def main():
    d, bj, oj, ot = sys.argv[1:5]
    hdr = sys.argv[5] if len(sys.argv) > 5 else ""
    bench = json.loads([ln for ln in open(bj) if ln.startswith("{")][-1])
    cols = bench["config"]["align_stats_per_step"]["columns"]
    tab = {c: per_kernel(d, c) for c in COUNTERS}
    kernels = sorted({k for c in tab.values() for k in c})
    # The pipeline steps of the profiled run are COUNTED from the trace (launches of align_fwd4_kernel / 2), not derived from the
    # bench line's flags: a run without --no-modes makes more steps than steps + warmup + 2 (round 5's file was normalised by 5
    # while the trace held 9: every per-step count 1.8 x too high).  The flags' count must agree when the header says the run was
    # --no-modes --no-coarse; anything else is refused.
    l4 = max([tab[c].get(k, (0.0, 0))[1] for c in COUNTERS for k in kernels if "align_fwd4_kernel" in k] + [0])
    if l4 <= 0 or l4 % FWD4_LAUNCHES_PER_STEP:
        raise SystemExit("pmc_counters: %d launches of align_fwd4_kernel in the trace: cannot count the pipeline steps" % l4)
    steps_run = l4 // FWD4_LAUNCHES_PER_STEP
    by_flags = bench["steps"] + bench["warmup"] + 2      # bench.py makes two residency calls before the warm-up
    if "--no-modes" in hdr and ("--no-coarse" in hdr or "--stage" in hdr) and by_flags != steps_run:
        raise SystemExit("pmc_counters: the trace holds %d pipeline steps, the bench line's flags say %d" % (steps_run, by_flags))
    res = {"_columns_per_step": cols, "_steps_in_profiled_run": steps_run, "_steps_by_bench_flags": by_flags, "_align_fwd4_launches": l4, "_command": hdr}
    lines = []
    for k in kernels:
        row = {c: tab[c].get(k, (0.0, 0))[0] for c in COUNTERS}
        launches = max(tab[c].get(k, (0.0, 0))[1] for c in COUNTERS)
        name = k
        for pat, st in STAGE.items():
            if pat.split("<")[0] in k and (("<" not in pat) or pat.split("<")[1].rstrip(">") + ">" in k.replace(" ", "").replace("(int)", "") or pat in k):
                name = st
        if not any(row.values()):
            continue
        e = res.setdefault(name, {"launches": 0, "valu_inst_per_step": 0.0, "salu_inst_per_step": 0.0, "wave_quad_cycles_per_step": 0.0,
                                  "active_valu_quad_cycles_per_step": 0.0})
        e["launches"] += launches
        e["valu_inst_per_step"] += row["SQ_INSTS_VALU"] / steps_run
        e["salu_inst_per_step"] += row["SQ_INSTS_SALU"] / steps_run
        e["wave_quad_cycles_per_step"] += row["SQ_WAVE_CYCLES"] / steps_run
        e["active_valu_quad_cycles_per_step"] += row["SQ_ACTIVE_INST_VALU"] / steps_run
        lines.append((k, launches, row))
    for name, e in res.items():
        if isinstance(e, dict) and name.startswith("align_"):
            e["valu_inst_per_pair_column"] = e["valu_inst_per_step"] / cols if cols else None
    json.dump(res, open(oj, "w"), indent=1, sort_keys=True)
    with open(ot, "w") as o:
        if hdr:
            o.write("# " + hdr + "\n")
        o.write("# sums over the launches of the run (%d pipeline steps, %d pair-columns per step); SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles\n" % (steps_run, cols))
        o.write("%-44s %8s " % ("kernel", "launches") + " ".join("%18s" % c for c in COUNTERS) + "\n")
        for k, launches, row in sorted(lines, key=lambda x: -x[2]["SQ_INSTS_VALU"])[:40]:
            o.write("%-44s %8d " % (k[:44], launches) + " ".join("%18d" % int(row[c]) for c in COUNTERS) + "\n")


if __name__ == "__main__":
    main()
