#!/usr/bin/env python3
"""Generates tests/golden/*: seeded synthetic inputs + the outputs of the REFERENCE ITSELF on them.

Run in the build container only (needs oracle/_ref/iteres, built from /root/reference by
`make -C oracle ref`).  What is committed is data: the input files we synthesise and the
output files the reference binary wrote.  No reference source text is stored.

    python tests/golden/make_golden.py            # regenerate everything
"""
from __future__ import annotations

import gzip
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from iteres_amd import synth  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
import refio  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "iteres")
KEEP_BIGWIG = {("quirks", "stat_default_bam")}
GZ_OVER = 150_000   # outputs/inputs larger than this are stored gzipped


def _store(src, dst):
    if os.path.getsize(src) > GZ_OVER:
        with open(src, "rb") as f, gzip.GzipFile(dst + ".gz", "wb", compresslevel=9, mtime=0) as g:
            shutil.copyfileobj(f, g)
    else:
        shutil.copyfile(src, dst)


def run_ref(case_dir, run_name, cmd, opts, aln, tmp, prefix="out"):
    """cmd: 'stat'|'filter'; opts: list of CLI options (without -o)."""
    out_dir = os.path.join(case_dir, run_name)
    os.makedirs(out_dir, exist_ok=True)
    work = os.path.join(tmp, run_name)
    os.makedirs(work, exist_ok=True)
    argv = [REF, cmd] + opts + ["-o", prefix, os.path.join(tmp, "chrom.sizes"), os.path.join(tmp, "rep.sizes"),
                                 os.path.join(tmp, "rmsk.txt"), os.path.join(tmp, aln)]
    pr = subprocess.run(argv, cwd=work, capture_output=True, text=True)
    files, bigwigs = [], {}
    for fn in sorted(os.listdir(work)):
        if fn.endswith(".bigWig"):
            # deflated by whatever zlib the reference was linked with: kept as a digest of the DECODED content
            # (tests/refio.py bigwig_digest); the bytes themselves only for the runs named in KEEP_BIGWIG
            raw = open(os.path.join(work, fn), "rb").read()
            bigwigs[fn] = refio.bigwig_digest(raw)
            if (os.path.basename(case_dir), run_name) in KEEP_BIGWIG:
                shutil.copyfile(os.path.join(work, fn), os.path.join(out_dir, fn))
            continue
        _store(os.path.join(work, fn), os.path.join(out_dir, fn))
        files.append(fn)
    return {"name": run_name, "cmd": cmd, "opts": opts, "aln": aln, "prefix": prefix, "rc": pr.returncode,
            "files": files, "bigwig_sha256": bigwigs, "stderr_tail": pr.stderr.replace("\r", "\n").strip().split("\n")[-3:]}


def emit_case(name, table, reads, runs, extra_rmsk_rows=(), extra_chrom_sizes=(), bam=True, sam=True, with_seq=True, extra_files=None):
    case_dir = os.path.join(HERE, name)
    if os.path.isdir(case_dir):
        shutil.rmtree(case_dir)
    os.makedirs(os.path.join(case_dir, "in"))
    with tempfile.TemporaryDirectory() as tmp:
        synth.write_sizes(os.path.join(tmp, "chrom.sizes"), list(table.chroms) + list(extra_chrom_sizes))
        synth.write_sizes(os.path.join(tmp, "rep.sizes"), table.rep_len.items())
        synth.write_rmsk(os.path.join(tmp, "rmsk.txt"), table, extra_rmsk_rows)
        alns = []
        if sam and reads is not None:
            synth.write_sam(os.path.join(tmp, "reads.sam"), reads, with_seq=with_seq)
            alns.append("reads.sam")
        if bam and reads is not None:
            synth.write_bam(os.path.join(tmp, "reads.bam"), reads, with_seq=with_seq)
            alns.append("reads.bam")
        for fn, text in (extra_files or {}).items():
            with open(os.path.join(tmp, fn), "w") as f:
                f.write(text)
            alns.append(fn)
        for fn in ["chrom.sizes", "rep.sizes", "rmsk.txt"] + alns:
            _store(os.path.join(tmp, fn), os.path.join(case_dir, "in", fn))
        man = {"case": name, "inputs": sorted(os.listdir(os.path.join(case_dir, "in"))), "runs": []}
        for rn, cmd, opts, aln in runs:
            man["runs"].append(run_ref(case_dir, rn, cmd, opts, aln, tmp))
        with open(os.path.join(case_dir, "manifest.json"), "w") as f:
            json.dump(man, f, indent=1)
    print(name, "->", len(man["runs"]), "runs")


# ------------------------------------------------------------------------------------------------ cases

def case_quirks():
    """Hand-written rows/reads, one per reference quirk."""
    chroms = [("chr1", 600000), ("chr2", 300000), ("chrS", 2)]
    names = ["AluY", "L1PA2", "MIRb", "NoLen", "(TG)n", "Tigger1"]
    fams = ["Alu", "L1", "MIR", "Simple_repeat", "TcMar-Tigger", "Odd"]
    clas = ["SINE", "LINE", "Simple_repeat", "DNA", "Other"]
    rep_len = {"AluY": 311, "L1PA2": 6100, "MIRb": 268, "(TG)n": 40, "Tigger1": 2418}
    rows = [
        # chrom,start,end,strand,name,fam,cla,cs,ce
        (0, 1000, 1300, "+", 0, 0, 0, 0, 300),          # plain AluY
        (0, 1250, 1500, "-", 2, 2, 0, 10, 260),         # overlaps the previous one by 50
        (0, 131000, 131200, "+", 0, 0, 0, 100, 311),     # spans the 128k bin boundary (131072) -> level-1 bin
        (0, 131060, 131100, "+", 4, 3, 2, 0, 40),        # level-0 bin 0, nested in the boundary spanner
        (0, 131080, 131400, "-", 1, 1, 1, 5000, 5320),   # level-0 bin 1
        (0, 5000, 11000, "+", 1, 1, 1, 0, 6000),        # long L1
        (0, 5200, 5300, "+", 3, 5, 4, 0, 100),          # NoLen: name without consensus length, nested in the L1
        (0, 20000, 20300, "+", 0, 5, 4, 20, 320),       # AluY row with ANOTHER family/class string (Odd/Other)
        (0, 0, 250, "+", 2, 2, 0, 0, 250),              # starts at 0: rend clamp against the GENOMIC end 250
        (0, 30000, 30100, "-", 5, 4, 3, 2400, 2500),    # cons_end beyond the consensus length (2418)
        (0, 30500, 30600, "+", 5, 4, 3, 900, 800),      # cons_end < cons_start
        (0, 599800, 600000, "+", 0, 0, 0, 0, 200),      # touches the chromosome end
        (1, 100, 400, "+", 1, 1, 1, 100, 400),
        (1, 100, 400, "-", 2, 2, 0, 0, 268),            # identical coordinates, other name: insertion order decides
        (1, 262100, 262200, "+", 0, 0, 0, 0, 100),      # spans 262144 (bin boundary) on chr2
    ]
    a = np.array
    t = synth.Table(chroms, a([r[0] for r in rows], np.int32), a([r[1] for r in rows], np.int64),
                    a([r[2] for r in rows], np.int64), a([ord(r[3]) for r in rows], np.uint8),
                    a([r[4] for r in rows], np.int32), a([r[5] for r in rows], np.int32), a([r[6] for r in rows], np.int32),
                    a([r[7] for r in rows], np.int64), a([r[8] for r in rows], np.int64), names, fams, clas, rep_len)
    extra_rows = [
        # chromosome absent from chrom.sizes: dropped at load (generic.c:1618-1622), still counted in the banner
        (585, 1, 0, 0, 0, "chrUn", 10, 200, -5, "+", "AluY", "SINE", "Alu", 0, 190, -121, 1),
        # comment line is skipped by lineFileNextRow
    ]
    header = [("chr1", 600000), ("chr2", 300000), ("chrS", 2), ("chrQ", 5000)]
    P, U, MU, RV, R1, R2 = synth.FPAIRED, synth.FUNMAP, synth.FMUNMAP, synth.FREVERSE, synth.FREAD1, synth.FREAD2
    recs = [
        # tid,pos,flag,mapq,cigar,mtid,mpos,isize
        (0, 1100, 0, 37, "50M", -1, -1, 0),             # inside AluY, '+': end = start+150 regardless of CIGAR
        (0, 1100, RV, 37, "50M", -1, -1, 0),            # '-': start = calend-150 = 1000
        (0, 1240, 0, 5, "50M", -1, -1, 0),              # MAPQ < Q; overlaps AluY (60) and MIRb (140): last increase wins
        (0, 900, 0, 60, "36M", -1, -1, 0),              # starts LEFT of AluY: counted, zero coverage (u32 wrap)
        (0, 131050, 0, 60, "100M", -1, -1, 0),          # three hits across bin levels: order coarse->fine, bin desc
        (0, 131090, RV, 20, "20M5D20M", -1, -1, 0),     # D advances calend
        (0, 131000, 0, 20, "10M2000N30M", -1, -1, 0),   # N advances calend ('+' ignores it because of -E)
        (0, 5250, RV, 60, "30=1X19M", -1, -1, 0),       # '=' / 'X' do NOT advance calend in samtools 0.1.18
        (0, 5150, 0, 60, "50M", -1, -1, 0),             # L1 vs nested NoLen
        (0, 20010, 0, 60, "50M", -1, -1, 0),            # AluY row with the odd family
        (0, 10, 0, 60, "50M", -1, -1, 0),               # repeat at 0
        (0, 60, RV, 60, "50M", -1, -1, 0),              # '-' with end < E -> start = 0
        (0, 30010, 0, 60, "50M", -1, -1, 0),            # cons_end beyond length
        (0, 30510, 0, 60, "50M", -1, -1, 0),            # cons_end < cons_start: no coverage
        (0, 599900, 0, 60, "80M", -1, -1, 0),           # end clipped to chromSize-1
        (0, 599990, 0, 60, "80M", -1, -1, 0),           # runs past the chromosome end
        (1, 150, 0, 60, "50M", -1, -1, 0),              # two rows with identical coordinates
        (1, 262120, RV, 60, "50M", -1, -1, 0),
        (1, 262000, 0, 60, "50M", -1, -1, 0),
        (2, 0, 0, 60, "1M", -1, -1, 0),                 # chromosome of size exactly 2 reads as "missing"
        (3, 100, 0, 60, "50M", -1, -1, 0),              # chromosome not in chrom.sizes
        (3, 200, 0, 60, "50M", -1, -1, 0),              # second read there: already in 'nochr'
        (0, 1000, U, 0, "*", -1, -1, 0),                # unmapped
        (0, 1100, P | R1, 60, "50M", 0, 1300, 250),      # proper pair, isize>0: [pos, pos+isize)
        (0, 1300, P | R2 | RV, 60, "50M", 0, 1100, -250),  # READ2: skipped (counted as end2)
        (0, 1300, P | R1 | RV, 60, "50M", 0, 1100, -250),  # READ1 reverse: [mpos, mpos-isize)
        (0, 1100, P | R1, 60, "50M", 0, 1900, 850),      # |isize| > I
        (0, 1100, P | R1, 60, "50M", 0, 1100, 0),        # isize == 0
        (0, 1120, P | R1 | MU, 60, "50M", -1, -1, 0),    # mate unmapped: SE-style unless -D
        (0, 1130, P | R2 | MU | RV, 3, "50M", -1, -1, 0),
        (0, 131075, 0, 60, "*", -1, -1, 0),             # mapped, no CIGAR (BAM only): end = pos + l_qseq
    ]
    n = len(recs)

    def cg(s):
        if s == "*":
            return []
        out, num = [], ""
        for ch in s:
            if ch.isdigit():
                num += ch
            else:
                out.append((ch, int(num)))
                num = ""
        return out
    cigs = [cg(r[4]) for r in recs]
    lq = [sum(l for op, l in c if op in "MIS=X") if c else 40 for c in cigs]
    r = synth.Reads(header, a([x[0] for x in recs], np.int32), a([x[1] for x in recs], np.int32),
                    a([x[2] for x in recs], np.uint16), a([x[3] for x in recs], np.uint8), a(lq, np.int32),
                    a([x[5] for x in recs], np.int32), a([x[6] for x in recs], np.int32), a([x[7] for x in recs], np.int32),
                    cigs, [f"q{i}" for i in range(n)])
    runs = [
        ("stat_default_bam", "stat", ["-w"], "reads.bam"),
        ("stat_default_sam", "stat", ["-w", "-S"], "reads.sam"),
        ("stat_E0", "stat", ["-w", "-E", "0"], "reads.bam"),
        ("stat_T", "stat", ["-w", "-T"], "reads.bam"),
        ("stat_D", "stat", ["-w", "-D"], "reads.bam"),
        ("stat_Q30_c05", "stat", ["-w", "-Q", "30", "-c", "0.5"], "reads.bam"),
        ("stat_N2_U1_I1000", "stat", ["-w", "-N", "2", "-U", "1", "-I", "1000"], "reads.bam"),
        ("stat_N1", "stat", ["-w", "-N", "1"], "reads.bam"),
        ("stat_N3_U2_x", "stat", ["-w", "-N", "3", "-U", "2", "-x"], "reads.bam"),
        ("filter_all", "filter", [], "reads.bam"),
        ("filter_all_r", "filter", ["-r"], "reads.bam"),
        ("filter_n_AluY", "filter", ["-n", "AluY", "-r"], "reads.bam"),
        ("filter_f_L1_t2", "filter", ["-f", "L1", "-t", "2"], "reads.bam"),
        ("filter_c_SINE_N2", "filter", ["-c", "SINE", "-N", "2", "-E", "0"], "reads.bam"),
        ("filter_T_g02", "filter", ["-T", "-g", "0.2", "-N", "3"], "reads.bam"),
    ]
    emit_case("quirks", t, r, runs, extra_rmsk_rows=extra_rows)


def case_mid():
    chroms = [("chrA", 3000000), ("chrB", 700000), ("chrC", 140000)]
    t = synth.make_table(101, chroms, 6000, n_names=300, n_fams=30, n_clas=9, overlap_frac=0.08, shuffle_frac=0.05,
                         inconsistent_frac=0.01)
    header = chroms + [("chrZ", 20000)]
    r = synth.make_reads(102, header, 20000, read_len=(30, 120), paired_frac=0.35, odd_cigar_frac=0.1)
    runs = [
        ("stat_default", "stat", ["-w"], "reads.bam"),
        ("stat_sam", "stat", ["-w", "-S"], "reads.sam"),
        ("stat_E0_Q20", "stat", ["-w", "-E", "0", "-Q", "20"], "reads.bam"),
        ("stat_T_c03", "stat", ["-w", "-T", "-c", "0.3"], "reads.bam"),
        ("stat_D_I300", "stat", ["-w", "-D", "-I", "300", "-E", "60"], "reads.bam"),
        ("filter_all", "filter", [], "reads.bam"),
        ("filter_n", "filter", ["-n", "Rep1", "-r"], "reads.bam"),
        ("filter_f", "filter", ["-f", t.fams[int(t.fam_of_row[0])], "-t", "3"], "reads.bam"),
    ]
    emit_case("mid", t, r, runs)


def case_manynames():
    """> 4096 and > 8192 distinct repNames: exercises the kent hash auto-resize that fixes output row order."""
    chroms = [("chr7", 9000000)]
    t = synth.make_table(201, chroms, 30000, n_names=9000, n_fams=70, n_clas=15, overlap_frac=0.02)
    r = synth.make_reads(202, chroms, 30000, read_len=(50, 50), odd_cigar_frac=0.0, nocigar_frac=0.0)
    runs = [("stat_default", "stat", [], "reads.bam")]     # no -w: stat files only (wig of 9000 names is large)
    emit_case("manynames", t, r, runs, sam=False, with_seq=False)


def case_cfg1():
    """BASELINE.json configs[0]: 100 k reads on a chr22-sized chromosome vs ~9e4 repeats."""
    chroms = [("chr22", 50818468)]
    t = synth.make_table(20260101, chroms, 90000, n_names=1200, n_fams=55, n_clas=18, overlap_frac=0.02)
    r = synth.make_reads(20260102, chroms, 100000, read_len=(100, 150), odd_cigar_frac=0.05, unmapped_frac=0.01)
    runs = [("stat_default", "stat", ["-w"], "reads.bam"), ("filter_all", "filter", [], "reads.bam")]
    emit_case("cfg1_chr22", t, r, runs, sam=False, with_seq=False)


def case_sidechan():
    """The host side channels: XA/NM tags (the multi-mapping veto, generic.c:303-341,972-982), exact duplicates (-R,
    generic.c:907-919) and the bed outputs (-B/-V, generic.c:925-936)."""
    chroms = [("chrA", 2000000), ("chrB", 500000)]
    t = synth.make_table(301, chroms, 2500, n_names=120, n_fams=20, n_clas=7, overlap_frac=0.06, shuffle_frac=0.03)
    header = chroms + [("chrZ", 20000)]
    r = synth.make_reads(302, header, 6000, read_len=(30, 120), paired_frac=0.2, odd_cigar_frac=0.05)
    rng = np.random.default_rng(303)
    n = len(r)
    # exact duplicates: a record repeats its predecessor's placement (same key chr:start:end:strand when both pass)
    for i in rng.choice(np.arange(1, n), 900, replace=False):
        for arr in (r.tid, r.pos, r.flag, r.l_qseq, r.mtid, r.mpos, r.isize):
            arr[i] = arr[i - 1]
        r.cigars[i] = list(r.cigars[i - 1])
    r.mapq[:8] = 60            # -R looks its key buffer up before ever writing it when the file starts with MAPQ < Q records
    # XA / NM on a third of the reads; alternatives land on repeats of the same name, of other names, on nothing,
    # and on chromosomes without repeats or absent from every file
    names = [nm for nm, _ in chroms]
    aux = [[] for _ in range(n)]
    for i in np.flatnonzero(rng.random(n) < 0.35):
        nm = int(rng.integers(0, 4))
        alts = []
        for _ in range(int(rng.integers(1, 5))):
            kind = rng.random()
            if kind < 0.65:
                row = int(rng.integers(0, len(t.start)))
                c, p0 = names[int(t.chrom[row])], int(t.start[row]) + int(rng.integers(-20, 40))
            elif kind < 0.85:
                ci = int(rng.integers(0, 2))
                c, p0 = names[ci], int(rng.integers(1, chroms[ci][1]))
            elif kind < 0.93:
                c, p0 = "chrZ", int(rng.integers(1, 20000))
            else:
                c, p0 = "chrNope", int(rng.integers(1, 5000))
            alts.append(f"{c},{'+' if rng.random() < 0.5 else '-'}{max(p0, 1)},{int(r.l_qseq[i])}M,{int(rng.integers(0, 5))}")
        fields = [f"XA:Z:{';'.join(alts)};"]
        if rng.random() < 0.9:
            fields.insert(0, f"NM:i:{nm}")
        if rng.random() < 0.3:
            fields.insert(0, "X0:i:1")
        aux[i] = fields
    r.aux = aux
    runs = [
        ("stat_veto", "stat", ["-w"], "reads.bam"),
        ("stat_x", "stat", ["-w", "-x"], "reads.bam"),
        ("stat_veto_sam", "stat", ["-w", "-S"], "reads.sam"),
        ("stat_R", "stat", ["-w", "-R"], "reads.bam"),
        ("stat_R_T_E0", "stat", ["-w", "-R", "-T", "-E", "0", "-x"], "reads.bam"),
        ("stat_B_V", "stat", ["-w", "-B", "-V"], "reads.bam"),
        ("stat_R_B_V_sam", "stat", ["-w", "-R", "-B", "-V", "-S", "-Q", "20"], "reads.sam"),
        ("filter_R", "filter", ["-R", "-r"], "reads.bam"),
        ("filter_xa_ignored", "filter", [], "reads.bam"),
    ]
    emit_case("sidechan", t, r, runs)


def case_cpg():
    """cpgstat / cpgfilter (generic.c:1064-1139): CpG sites with scores from a bedGraph file; the sums are doubles added
    in file order, the lookup takes the FIRST row binKeeperFind returns."""
    chroms = [("chrA", 3000000), ("chrB", 700000), ("chrC", 140000)]
    t = synth.make_table(401, chroms, 5000, n_names=200, n_fams=25, n_clas=8, overlap_frac=0.12, shuffle_frac=0.05)
    rng = np.random.default_rng(402)
    lines = ["# CpG methylation scores", ""]
    n = 40000
    ci = rng.choice(3, n, p=[0.75, 0.2, 0.05])
    pos = (rng.random(n) * np.array([c[1] for c in chroms])[ci]).astype(np.int64)
    order = np.lexsort((pos, ci))
    order[5000:5400] = order[5000:5400][::-1]                      # a stretch out of order: the sums follow the file
    for k in order:
        c, p0 = chroms[ci[k]][0], int(pos[k])
        score = float(rng.choice([rng.random() * 3, rng.random() * 1e-3, 10 ** rng.uniform(-6, 2), 0.1, 1 / 3]))
        end = p0 + 2 if rng.random() < 0.97 else p0 + int(rng.integers(1, 400))
        lines.append(f"{c}\t{p0}\t{end}\t{score!r}")
    lines[700] = "   # an indented comment"
    lines.insert(900, "   ")
    lines.append("chrNoRepeats\t100\t102\t1.5")                    # a chromosome without rows
    lines.append(f"chrA\t{chroms[0][1] - 1}\t{chroms[0][1] + 5}\t2.25\textra\tfields")
    lines.append("chrB 10 12 0.75")                                  # blanks separate fields too
    bed = "\n".join(lines) + "\n"
    fam = t.fams[int(t.fam_of_row[7])]
    cla = t.clas[int(t.cla_of_row[11])]
    runs = [
        ("cpgstat_w", "cpgstat", ["-w"], "cpg.bedGraph"),
        ("cpgstat", "cpgstat", [], "cpg.bedGraph"),
        ("cpgfilter_all", "cpgfilter", [], "cpg.bedGraph"),
        ("cpgfilter_n_t", "cpgfilter", ["-n", "Rep3", "-t", "0.5"], "cpg.bedGraph"),
        ("cpgfilter_f", "cpgfilter", ["-f", fam], "cpg.bedGraph"),
        ("cpgfilter_c_t", "cpgfilter", ["-c", cla, "-t", "2"], "cpg.bedGraph"),
    ]
    emit_case("cpg", t, None, runs, bam=False, sam=False, extra_files={"cpg.bedGraph": bed})


def case_addchr():
    """-C (generic.c:781-791): header names without the "chr" prefix get it, "MT" becomes chrM, "GL*" references are
    skipped before the unknown-chromosome test, names that already start with "chr" stay; without -C none of the short
    names is in the size file (one warning each)."""
    chroms = [("chr1", 900000), ("chr2", 500000), ("chrM", 16571), ("chrX", 300000)]
    t = synth.make_table(501, chroms, 3000, n_names=150, n_fams=20, n_clas=7, overlap_frac=0.05)
    header = [("1", 900000), ("2", 500000), ("MT", 16571), ("chrX", 300000), ("GL000191.1", 106433), ("mt", 16571), ("HSCHR6_MHC", 50000)]
    r = synth.make_reads(502, header, 12000, read_len=(40, 110), paired_frac=0.25, odd_cigar_frac=0.05)
    runs = [
        ("stat_C", "stat", ["-w", "-C"], "reads.bam"),
        ("stat_C_sam", "stat", ["-w", "-C", "-S"], "reads.sam"),
        ("stat_noC", "stat", ["-w"], "reads.bam"),
        ("filter_C", "filter", ["-C", "-r"], "reads.bam"),
        ("stat_C_R_B", "stat", ["-w", "-C", "-R", "-B", "-V"], "reads.bam"),
    ]
    emit_case("addchr", t, r, runs)


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("build the reference first: make -C oracle ref")
    which = sys.argv[1:] or ["quirks", "mid", "manynames", "cfg1", "sidechan", "cpg", "addchr"]
    for w in which:
        globals()["case_" + w]()
