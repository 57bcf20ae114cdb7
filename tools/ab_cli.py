#!/usr/bin/env python3
"""Scratch A/B of the whole command on one box: python tools/ab_cli.py <n_reads> <seq_len> <reps> NAME:VAR=VAL,VAR=VAL ...
Generates the bench inputs once (bench.py's generator), then runs `iteres stat -w` <reps> times per setting, interleaved.
LD_LIBRARY_PATH in a setting selects another build of libiteres_amd.so (the program's RUNPATH comes after it)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class A:
    pass


def main():
    a = A()
    a.reads, a.seq_len, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    a.rows, a.cpu_reads, a.workdir = 5_500_000, 0, ""
    # ITX_AB_MKBAM="content=hiseq cigar=mixed paired=1 pileup=20": what the BAM looks like (default: round 2's legacy content)
    for kv in os.environ.get("ITX_AB_MKBAM", "").split():
        k, v = kv.split("=", 1)
        setattr(a, k, int(v) if v.isdigit() else v)
    wd, info = bench.ensure_inputs(a, 16)
    settings = [("base", {})]
    for spec in sys.argv[4:]:
        name, kv = spec.split(":", 1)
        settings.append((name, dict(x.split("=", 1) for x in kv.split(",") if x)))
    walls = {n: [] for n, _ in settings}
    scans = {n: [] for n, _ in settings}
    notes = {}
    for rep in range(reps):
        for name, kv in settings:
            env = dict(os.environ, OMP_NUM_THREADS="16", ITX_TIMING="1", ITX_GPUS="1")
            for k, v in kv.items():
                env[k] = os.path.join(ROOT, v) if k == "LD_LIBRARY_PATH" else v
            out = os.path.join(wd, "ab_" + name)
            args = bench.base_args(wd)
            extra = os.environ.get("ITX_AB_OPTS", "").split()              # e.g. ITX_AB_OPTS="-R": options behind `stat -w`
            args = args[:2] + extra + args[2:]
            wall, rc, err, seen = bench.run_timed(bench.OURS, args + [os.path.join(wd, "reads.bam")], out, env, (bench.SCAN_BEGIN, bench.SCAN_END))
            assert rc == 0, err[-800:]
            walls[name].append(round(wall, 3))
            scans[name].append(round(seen.get(bench.SCAN_END, 0) - seen.get(bench.SCAN_BEGIN, 0), 3))
            notes[name] = [ln[13:] for ln in err.replace("\r", "\n").split("\n") if ln.startswith("[itx timing]")]
    same = {}
    for name, _ in settings[1:]:
        same[name] = all(open(os.path.join(wd, "ab_base", fn), "rb").read() == open(os.path.join(wd, "ab_" + name, fn), "rb").read() for fn in bench.TEXT_OUTPUTS)
    print(json.dumps({"reads": a.reads, "seq_len": a.seq_len, "walls_s": walls, "scan_s": scans, "same_outputs_as_base": same, "notes": notes, "inputs": info}), flush=True)


if __name__ == "__main__":
    main()
