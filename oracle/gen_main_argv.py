#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY (build container; never on the GPU box).  Writes tests/golden/main_argv.json: the command lines the
REFERENCE's own driver builds for the stages of the dynamic-boundary path -- /root/reference/main.py:479-482 (split_genome_chunks.py),
:520-532 (coarse_boundary.py), :545-562 (judge_TIR_transposons.py), :580-594 (judge_Helitron_transposons.py), :612-628
(judge_Non_LTR_transposons.py).  main.py is RUN (runpy, its __main__ block) on a miniature genome with `os.system` replaced by a
recorder: no stage executes; the recorder only leaves behind the files main.py looks for next (genome.fa.clean, genome.cut<i>.fa).
The fixture holds the argv lists with the run's temporary directory written as {OUT}; nothing of main.py's text is stored.

    python oracle/gen_main_argv.py
"""
import json
import os
import runpy
import shlex
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_harness  # noqa: E402

STAGES = ("split_genome_chunks.py", "coarse_boundary.py", "judge_TIR_transposons.py", "judge_Helitron_transposons.py", "judge_Non_LTR_transposons.py")


def run_main(extra_args, tag):
    import numpy as np

    import casegen

    ref_harness.load_reference_util()
    if ref_harness.REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, ref_harness.REFERENCE_ROOT)
    tmp = tempfile.mkdtemp(prefix="hite_main_argv_")
    genome = os.path.join(tmp, "in", "genome.fa")
    os.makedirs(os.path.dirname(genome))
    rng = np.random.default_rng(520)
    with open(genome, "w") as f:
        for c in range(2):
            f.write(">chr%d\n%s\n" % (c + 1, casegen.rand_seq(rng, 30_000)))
    out = os.path.join(tmp, "out")
    recorded = []

    def fake_system(cmd):
        parts = shlex.split(cmd.split(">")[0])
        if not parts:
            return 0
        prog = os.path.basename(parts[0])
        if prog in STAGES:
            recorded.append([prog] + parts[1:])
        if prog == "genome_clean.py":
            shutil.copyfile(parts[parts.index("-i") + 1], parts[parts.index("-o") + 1])
        elif prog == "split_genome_chunks.py":
            for i in range(2):      # two chunks: the loop of step 3 runs twice, ref_index 0 and 1
                shutil.copyfile(parts[parts.index("-g") + 1], os.path.join(out, "genome.cut%d.fa" % i))
        elif prog == "touch":
            for p_ in parts[1:]:
                if os.path.abspath(p_).startswith(tmp):
                    open(p_, "a").close()
        return 0

    saved = (os.system, sys.argv, os.getcwd())
    os.system = fake_system
    sys.argv = ["main.py", "--genome", genome, "--out_dir", out, "--work_dir", os.path.join(tmp, "work"), "--thread", "12", "--annotate", "0"] + extra_args
    try:
        try:
            runpy.run_path(os.path.join(ref_harness.REFERENCE_ROOT, "main.py"), run_name="__main__")
        except (SystemExit, Exception) as e:          # what comes after step 3 needs the stages' outputs; the argv are recorded by then
            print("main.py stopped after %d recorded stage commands: %s: %s" % (len(recorded), type(e).__name__, e))
    finally:
        os.system, sys.argv = saved[0], saved[1]
        os.chdir(saved[2])
    sub = lambda a: a.replace(out, "{OUT}").replace(os.path.join(tmp, "work"), "{WORK}").replace(tmp, "{TMP}")  # noqa: E731
    # (the chunk files come out of os.listdir in file-system order: the fixture keeps them by --ref_index, stages in main.py's order)
    def key(c):
        return (int(c[c.index("--ref_index") + 1]) if "--ref_index" in c else -1, STAGES.index(c[0]))
    recorded.sort(key=key)
    res = dict(tag=tag, extra_args=extra_args, threads_given=12, commands=[[sub(a) for a in c] for c in recorded])
    shutil.rmtree(tmp, ignore_errors=True)
    return res


def main():
    runs = [run_main([], "defaults"), run_main(["--plant", "0", "--te_type", "tir", "--debug", "1", "--recover", "0", "--flanking_len", "50"], "animal_tir_debug")]
    for r in runs:
        print(r["tag"], [c[0] for c in r["commands"]])
        assert any(c[0] == "coarse_boundary.py" for c in r["commands"])
    path = os.path.join(ROOT, "tests", "golden", "main_argv.json")
    with open(path, "w") as f:
        json.dump(runs, f, indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
