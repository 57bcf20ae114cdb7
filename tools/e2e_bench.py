#!/usr/bin/env python3
"""End-to-end run on FILES (BAM in, .stat/.wig out): the drop-in CLI next to the reference binary on the same inputs.

    python tools/e2e_bench.py [n_reads=50000000] [n_rows=5500000] [seq_len=0] [ref_reads=n_reads]

Generates the hg38-scale synthetic rmsk / size files (Python) and a coordinate-sorted BAM (tools/mkbam.c), runs
`iteres stat -w` with both programs, prints one JSON line with wall times, M reads/s, and whether the output files
are byte-identical. The reference binary is baseline and checker here (test infrastructure)."""
import filecmp
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iteres_amd import synth  # noqa: E402


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    n_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 5_500_000
    seq_len = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    ref_reads = int(sys.argv[4]) if len(sys.argv) > 4 else n_reads
    threads = os.environ.get("OMP_NUM_THREADS", "16")
    ref = os.path.join(ROOT, "oracle", "_ref", "iteres")
    ours = os.path.join(ROOT, "iteres_amd", "host", "iteres")
    mk = os.path.join(ROOT, "tools", "mkbam")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-o", mk, os.path.join(ROOT, "tools", "mkbam.c"), "-lz", "-ldl"])
    scale = n_rows / 5_500_000
    chroms = synth.HG38_CHROMS if scale == 1 else [(n, max(int(s * scale), 1000)) for n, s in synth.HG38_CHROMS]
    tmp = tempfile.mkdtemp(prefix="itx_e2e_", dir=os.environ.get("ITX_TMP", None))
    t0 = time.time()
    tb = synth.make_table(20260101, chroms, n_rows, n_names=15000, n_fams=60, n_clas=20, overlap_frac=0.02)
    synth.write_sizes(os.path.join(tmp, "chrom.sizes"), chroms)
    synth.write_sizes(os.path.join(tmp, "rep.sizes"), tb.rep_len.items())
    synth.write_rmsk(os.path.join(tmp, "rmsk.txt"), tb)
    env = dict(os.environ, OMP_NUM_THREADS=threads, ITX_TIMING="1")
    subprocess.check_call([mk, os.path.join(tmp, "chrom.sizes"), str(n_reads), os.path.join(tmp, "reads.bam"), str(seq_len), "7"], env=env)
    same_bam = ref_reads == n_reads
    if not same_bam:
        subprocess.check_call([mk, os.path.join(tmp, "chrom.sizes"), str(ref_reads), os.path.join(tmp, "reads_ref.bam"), str(seq_len), "7"], env=env)
    out = {"n_reads": n_reads, "n_rows": n_rows, "seq_len": seq_len, "bam_MB": round(os.path.getsize(os.path.join(tmp, "reads.bam")) / 1e6, 1),
           "gen_s": round(time.time() - t0, 1), "host_threads": int(threads), "host_cores": os.cpu_count()}
    # ITX_E2E_CMD: another command line head than `stat -w`, e.g. "filter -n Rep1" (BASELINE configs[4])
    head = os.environ.get("ITX_E2E_CMD", "stat -w").split()
    out["command"] = " ".join(head)
    base = head + ["-o", "out", os.path.join(tmp, "chrom.sizes"), os.path.join(tmp, "rep.sizes"), os.path.join(tmp, "rmsk.txt")]
    runs = [("drop_in", ours, "reads.bam", n_reads)]
    if os.path.exists(ref) and ref_reads > 0:
        runs.append(("reference", ref, "reads.bam" if same_bam else "reads_ref.bam", ref_reads))
    for name, exe, bam, nr in runs:
        wd = os.path.join(tmp, name)
        os.makedirs(wd)
        t1 = time.time()
        pr = subprocess.run([exe] + base + [os.path.join(tmp, bam)], cwd=wd, capture_output=True, text=True, env=env)
        dt = time.time() - t1
        out[name] = {"rc": pr.returncode, "reads": nr, "wall_s": round(dt, 2), "M_reads_per_s_whole_command": round(nr / dt / 1e6, 3)}
        if pr.returncode != 0:
            out[name]["stderr_tail"] = pr.stderr[-400:]
        tl = [l for l in pr.stderr.split("\n") if l.startswith("[itx timing]")]
        if tl:
            out[name]["phases"] = tl
    # A/B inside one box: ITX_E2E_AB="VAR=1" runs the drop-in three more times each without and with that setting
    ab = os.environ.get("ITX_E2E_AB")
    if ab:
        k, v = ab.split("=", 1)
        walls = {"base": [], ab: []}
        for rep in range(int(os.environ.get("ITX_E2E_AB_REPS", "3"))):
            for tag, e in (("base", env), (ab, dict(env, **{k: v}))):
                wd = os.path.join(tmp, f"ab_{tag}_{rep}".replace("=", "_"))
                os.makedirs(wd)
                t1 = time.time()
                subprocess.run([ours] + base + [os.path.join(tmp, "reads.bam")], cwd=wd, capture_output=True, text=True, env=e)
                walls[tag].append(round(time.time() - t1, 3))
        out["ab_walls_s"] = walls
    # ITX_E2E_ROCPROF=<dir>: the drop-in once more under rocprofv3 (kernel trace + stats) for the per-kernel times of a whole run
    prof = os.environ.get("ITX_E2E_ROCPROF")
    if prof:
        wd = os.path.join(tmp, "prof")
        os.makedirs(wd)
        os.makedirs(prof, exist_ok=True)
        subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.abspath(prof), "-o", "cli", "--", ours] + base
                       + [os.path.join(tmp, "reads.bam")], cwd=wd, capture_output=True, text=True, env=env)
    if "reference" in out and same_bam:
        same = {}
        for fn in sorted(os.listdir(os.path.join(tmp, "reference"))):
            if fn.endswith(".bigWig"):
                continue
            a, b = os.path.join(tmp, "reference", fn), os.path.join(tmp, "drop_in", fn)
            same[fn] = os.path.exists(b) and filecmp.cmp(a, b, shallow=False)
        out["files_identical"] = same
        # the two bigWigs by DECODED content (chromosomes, sections, zoom records, summary: tests/refio.py); their bytes depend on the zlib at hand
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import refio
        bw = {}
        for fn in sorted(os.listdir(os.path.join(tmp, "reference"))):
            if fn.endswith(".bigWig"):
                a, b = os.path.join(tmp, "reference", fn), os.path.join(tmp, "drop_in", fn)
                bw[fn] = os.path.exists(b) and refio.bigwig_digest(open(a, "rb").read()) == refio.bigwig_digest(open(b, "rb").read())
        out["bigwig_content_identical"] = bw
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
