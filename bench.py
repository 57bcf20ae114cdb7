#!/usr/bin/env python3
"""bench.py — M alignments/s through `iteres stat` (hg38 rmsk) on MI355X: the drop-in command, end to end, on files.

    python bench.py --gpus N --steps K --warmup W
    (N > 1 without WORLD_SIZE in the environment: bench.py starts `python -m torch.distributed.run` with N ranks itself,
     as a child process, before anything touches the GPU; under torch.distributed.run it is one rank per GPU.)

Workload, N = 1 (BASELINE.json configs[2]): a coordinate-sorted 500 M-read synthetic hg38 BAM whose records carry 100 bases
+ qualities (tools/mkbam.c, BGZF level 1) with the CONTENT of a sequencer's BAM — `--content hiseq` (the default): independent
bases, 40-value qualities, Illumina-style names, 5 % CIGARs with S / D / I / N: ~112 B/read compressed, ~220 B/read inflated,
23 k literals + 7 k matches per BGZF block (stated in `config.content`, counted from the file) — `--content legacy` is round 2's
file, 97 % of whose bytes come out of LZ77 matches — against a 5.5 M-row
RepeatMasker-like table (15 k names / 60 families / 20 classes), `iteres stat -w` with the reference's defaults
(-Q 10 -E 150 -c 1e-4), per-base coverage wigs kept. One "step" = ONE whole run of the command
(iteres_amd/host/iteres: size files + rmsk parse -> table build -> BAM decode on the device -> overlap classification +
accumulation -> .stat/.wig/.bigWig/.report written): `value` = reads / wall time of the command — the clock the
reference itself prints (stat.c:45,183-184). `scan_only` is the banner-to-banner scan phase (stat.c:144,153).

N > 1 (configs[3]): weak scaling — the command is given a file list of N copies (hard links) of that BAM, its ranks (one
process per GPU) take equal shares of the list's compressed bytes, every rank holds a replica of the table and its own
accumulators, and ONE sum-reduce (RCCL over xGMI) of the compact partial onto rank 0 ends the stream; rank 0 writes the
files. `strong_scaling` in the same line: the ONE 500 M-read BAM split N ways.

`filter_leg` (configs[4]): `iteres filter -c <the biggest class>` end to end on the same BAM, the reference on the sample beside
it, and the HIP-event time of k_stream<ATOMIC_LOCUS> on the resident records. `cpu_baseline_mt`: our CPU restatement (the
oracle) on the 16 cores of the box's CPU share — the hot path only.

`roofline`: the overlap kernel k_stream (derive + classify + key emit) on the same number of records RESIDENT in HBM,
timed with HIP events on the submitting stream around every launch; bytes counted both ways (SURVEY.md §8(d) K1 bytes
= `achieved`/`frac`; the as-built movement = `as_built`). `cpu_baseline` (rank 0, N = 1): the REFERENCE binary
(oracle/_ref/iteres, compiled from the reference's own sources — test infrastructure) on a bounded prefix-sized sample
of the same workload (median of 3 runs; the same sources at -O2 beside it); every text output of the two programs on that sample
is compared byte for byte.
"""
from __future__ import annotations

import argparse
import filecmp
import hashlib
import json
import os
import shutil
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
OURS = os.path.join(ROOT, "iteres_amd", "host", "iteres")
REF = os.path.join(ROOT, "oracle", "_ref", "iteres")
REF_O2 = os.path.join(ROOT, "oracle", "_ref", "o2", "iteres")
MKBAM = os.path.join(ROOT, "tools", "mkbam")
TEXT_OUTPUTS = ("out.iteres.subfamily.stat", "out.iteres.family.stat", "out.iteres.class.stat", "out.iteres.report", "out.iteres.wig",
                "out.iteres.unique.wig")


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=500_000_000, help="reads in the BAM (configs[2]: 500 M)")
    ap.add_argument("--seq-len", type=int, default=100, help="bases + qualities per record in the BAM (0: none)")
    ap.add_argument("--content", default="hiseq", choices=("legacy", "hiseq", "novaseq"),
                    help="what the BAM's SEQ/QUAL/names look like (tools/mkbam.c): hiseq = independent bases + 40-value qualities (literal-heavy, the default), "
                         "novaseq = 4-bin qualities, legacy = round 2's file (97 %% of the bytes out of LZ77 matches)")
    ap.add_argument("--cigar", default="mixed", choices=("simple", "mixed"), help="mixed: 5 %% of the CIGARs carry S / D / I / N (SURVEY.md 8(d))")
    ap.add_argument("--paired", type=int, default=0, help="1: fragments of two reads, isize ~ N(350, 60) (SURVEY.md 8(d)'s paired variant)")
    ap.add_argument("--pileup", type=int, default=0, help="K loci of the genome at ~2000x depth")
    ap.add_argument("--rows", type=int, default=5_500_000)
    ap.add_argument("--cpu-reads", type=int, default=10_000_000, help="reads of the sample the reference binary is timed on (0 = skip)")
    ap.add_argument("--cpu-repeats", type=int, default=3, help="runs of the reference binary on the sample (the median is reported)")
    ap.add_argument("--filter-steps", type=int, default=2, help="timed runs of `iteres filter -c <class>` on the same BAM (BASELINE configs[4]; 0 = skip)")
    ap.add_argument("--replay-steps", type=int, default=10, help="launches of the resident hot path behind `roofline` (0 = skip)")
    ap.add_argument("--replay-reads", type=int, default=0, help="records resident for the roofline replay (0 = --reads)")
    ap.add_argument("--threads", type=int, default=0, help="host threads of the command (0: min(16, cores))")
    ap.add_argument("--workdir", default=os.environ.get("ITX_BENCH_DIR", ""))
    ap.add_argument("--keep", action="store_true", help="keep the generated inputs (they are reused when present)")
    ap.add_argument("--settle", action="store_true", help="take and release most of the card once before anything is timed (see vram_settle; round 2's default, "
                    "when the command reserved 50 GB per run — it reserves 12 GB now and the warm-up runs see to those)")
    ap.add_argument("--no-settle", action="store_true", help="(the default now; kept for old command lines)")
    ap.add_argument("--no-replay-check", action="store_true", help="profiling runs: skip the small oracle-checked launch, so that every k_stream launch of this process has the replay's size")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ device memory
SETTLE_CHILD = r'''
import os, sys, time, torch
dev = int(sys.argv[1]); want = float(sys.argv[2])
torch.cuda.set_device(dev)
free, total = torch.cuda.mem_get_info()
goal = min(want * (1 << 30), 0.8 * free)
keep, got, t0, slow = [], 0, time.perf_counter(), 0.0
while got < goal:
    t1 = time.perf_counter()
    keep.append(torch.empty(8 << 30, dtype=torch.uint8, device="cuda")); torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    slow += dt if dt > 0.05 else 0.0
    got += 8 << 30
print(f"{got / (1 << 30):.0f} {time.perf_counter() - t0:.3f} {slow:.3f}", flush=True)
os._exit(0)
'''


def vram_settle(device, gb=176.0, pause=6.0):
    """A fresh box hands out device memory the driver has yet to clear: the first allocations of tens of GB then take
    1.3 - 2.2 s per 50 GB (tools/release_cost.py; memory a process released is cleared in the background and comes back at
    once). The command allocates ~50 GB per run, so on such a box the first runs lose seconds inside hipMalloc — box state,
    not the command's. Before anything is timed (and before the minutes of input generation) a child process takes and
    releases most of the card once; what it found is reported in the JSON line (`vram_settle`)."""
    t0 = time.perf_counter()
    try:
        out = subprocess.run([sys.executable, "-c", SETTLE_CHILD, str(device), str(gb)], capture_output=True, text=True, timeout=120)
        f = out.stdout.split()
        res = {"allocated_GB": float(f[0]), "seconds": float(f[1]), "seconds_in_slow_allocations": float(f[2])}
    except Exception as e:                                                    # never fatal: it is only a warm-up
        res = {"error": str(e)[:200]}
    if res.get("seconds_in_slow_allocations", 0) > 0.5:
        time.sleep(pause)                                                     # the release is cleared in the background
    res["total_s"] = round(time.perf_counter() - t0, 2)
    res["what"] = "one child process allocated and released this much device memory before inputs, warm-up and timed steps (bench.py vram_settle)"
    return res


# ------------------------------------------------------------------------------------------------ inputs (rank 0)
def ensure_inputs(a, threads):
    """chrom.sizes / rep.sizes / rmsk.txt / reads.bam / sample.bam for these parameters, generated once per box."""
    from iteres_amd import synth
    mkopts = mkbam_options(a)
    key = f"r{a.reads}_s{a.seq_len}_t{a.rows}_c{a.cpu_reads if getattr(a, 'gpus', 1) == 1 else 0}_" + "_".join(o.split("=")[1] for o in mkopts)
    wd = a.workdir or os.path.join("/tmp", f"itx_bench_{key}")
    os.makedirs(wd, exist_ok=True)
    done = os.path.join(wd, "inputs.json")
    if os.path.exists(done):
        return wd, json.load(open(done))
    t0 = time.time()
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-o", MKBAM, os.path.join(ROOT, "tools", "mkbam.c"), "-lz", "-ldl"])
    scale = a.rows / 5_500_000
    chroms = synth.HG38_CHROMS if scale == 1 else [(n, max(int(s * scale), 1000)) for n, s in synth.HG38_CHROMS]
    tb = synth.make_table(20260101, chroms, a.rows, n_names=15000, n_fams=60, n_clas=20, overlap_frac=0.02)
    synth.write_sizes(os.path.join(wd, "chrom.sizes"), chroms)
    synth.write_sizes(os.path.join(wd, "rep.sizes"), tb.rep_len.items())
    synth.write_rmsk(os.path.join(wd, "rmsk.txt"), tb, workers=threads)
    t1 = time.time()
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    subprocess.check_call([MKBAM, os.path.join(wd, "chrom.sizes"), str(a.reads), os.path.join(wd, "reads.bam"), str(a.seq_len), "7"] + mkopts, env=env)
    if a.cpu_reads > 0 and getattr(a, 'gpus', 1) == 1:
        subprocess.check_call([MKBAM, os.path.join(wd, "chrom.sizes"), str(a.cpu_reads), os.path.join(wd, "sample.bam"), str(a.seq_len), "7"] + mkopts, env=env)
    info = {"table_s": round(t1 - t0, 1), "bam_s": round(time.time() - t1, 1), "bam_bytes": os.path.getsize(os.path.join(wd, "reads.bam")),
            "n_rep": len(tb.names), "n_fam": len(tb.fams), "n_cla": len(tb.clas)}
    # what the decoder has to chew: literals / matches per BGZF block, bytes per read (a sample of blocks spread over the file,
    # counted by pass 1 of the product's decoder built for the host and compared with zlib)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bam_content
        info["content"] = bam_content.content_stats(os.path.join(wd, "reads.bam"), 200)
    except Exception as e:                                                    # never fatal: it is a description
        info["content"] = {"error": str(e)[:200]}
    # the filter leg (configs[4]) takes the biggest class: how many of the rows it holds
    import numpy as np
    per_cla = np.bincount(np.asarray(tb.cla_of_row), minlength=len(tb.clas))
    k = int(per_cla.argmax())
    info["filter_class"] = {"name": tb.clas[k], "rows": int(per_cla[k]), "fraction_of_rows": round(float(per_cla[k]) / max(len(tb.cla_of_row), 1), 4)}
    json.dump(info, open(done, "w"))
    return wd, info


def mkbam_options(a):
    o = [f"content={getattr(a, 'content', 'legacy')}", f"cigar={getattr(a, 'cigar', 'simple')}"]
    if getattr(a, "paired", 0):
        o.append("paired=1")
    if getattr(a, "pileup", 0):
        o.append(f"pileup={a.pileup}")
    return o


def base_args(wd):
    return ["stat", "-w", "-o", "out", os.path.join(wd, "chrom.sizes"), os.path.join(wd, "rep.sizes"), os.path.join(wd, "rmsk.txt")]


def run_timed(exe, args, cwd, env, banners=()):
    """Runs a command; returns (wall seconds, return code, stderr text, {banner: seconds after start when it appeared})."""
    os.makedirs(cwd, exist_ok=True)
    t0 = time.perf_counter()
    p = subprocess.Popen([exe] + args, cwd=cwd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    seen, chunks, tail = {}, [], b""
    fd = p.stderr.fileno()
    while True:
        b = os.read(fd, 1 << 16)
        if not b:
            break
        now = time.perf_counter() - t0
        chunks.append(b)
        window = tail + b
        for s in banners:
            if s not in seen and s.encode() in window:
                seen[s] = now
        tail = window[-128:]
    rc = p.wait()
    wall = time.perf_counter() - t0
    return wall, rc, b"".join(chunks).decode("utf-8", "replace"), seen


SCAN_BEGIN, SCAN_END = "* Parsing the SAM/BAM file", "* Writing stats and Wig file"


def report_counts(path):
    """numbers of the .report file (generic.c:53-70), in file order"""
    out = []
    with open(path) as f:
        for ln in f:
            parts = ln.replace(":", " ").replace("\t", " ").split()
            for tok in parts[::-1]:
                if tok.isdigit():
                    out.append(int(tok))
                    break
    return out


# ------------------------------------------------------------------------------------------------ resident replay
def resident_roofline(a, n_records, steps, rank, with_check):
    """The hot path on n_records records resident in HBM (seeded on the device): HIP-event time of every k_stream launch."""
    import numpy as np
    import torch
    from iteres_amd import engine as eng, synth
    dev = torch.device("cuda", torch.cuda.current_device())
    scale = a.rows / 5_500_000
    chroms = synth.HG38_CHROMS if scale == 1 else [(n, max(int(s * scale), 1000)) for n, s in synth.HG38_CHROMS]
    tb = synth.make_table(20260101, chroms, a.rows, n_names=15000, n_fams=60, n_clas=20, overlap_frac=0.02)
    rep_len = np.array([tb.rep_len.get(n, 0) for n in tb.names], np.uint32)
    rows = eng.make_rows(tb.chrom, tb.start, tb.end, tb.cons_start, tb.cons_end, tb.rep_name, tb.fam_of_row, tb.cla_of_row)
    cs = np.array([s for _, s in chroms], np.int64)
    table = eng.Table(rows, cs, rep_len, len(tb.fams), len(tb.clas), device=dev.index)
    e = eng.Engine(table, {}, batch_capacity=n_records)
    e.set_tidmap(list(range(len(chroms))))
    d = synth.make_reads_device(20260102 + rank, chroms, n_records, dev)      # the records, made in HBM
    ptrs = {k: v.data_ptr() for k, v in d.items()}
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        e.submit_device(ptrs, n_records, stream=stream)
    e.sync()
    e.reset()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        e.submit_device(ptrs, n_records, stream=stream)
    e.sync()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t1
    st = e.stats()
    res = e.finish()
    subs = max(st["submits"], 1)
    stream_ms = st["stage_ms"][0] / subs
    keys = st["keys"]                                  # of the last launch (every launch sees the same records)
    n_rows = int(table.info.n_rows)
    index_bytes = int(sum(s for _, s in chroms) >> int(table.info.bin_shift)) * 8
    # 14 B in + 4 B out per read, the table (28 B x rows) and its index once per launch
    survey_bytes = 18 * n_records + 28 * n_rows + index_bytes
    # what the kernel moves as built: 14 B per record in, 8 B per emitted key out, 32-byte rows and the binned index once
    built_bytes = 14 * n_records + 8 * keys + 32 * n_rows + index_bytes
    sec = stream_ms * 1e-3
    out = {"bound": "hbm", "kernel": "k_stream<EMIT> (derive + classify + key emit + partition count)",
           "achieved": round(survey_bytes / sec / 1e9, 2) if sec > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(survey_bytes / sec / 1e9 / HBM_PEAK_GBS, 4) if sec > 0 else 0.0,
           "bytes_accounting": "SURVEY.md 8(d) K1: 18 B/read + 28 B x table rows + index, per launch",
           "algorithmic_bytes_per_launch": int(survey_bytes), "avg_launch_ms": round(stream_ms, 4), "launches": int(subs),
           "records_per_launch": n_records,
           "as_built": {"bytes_per_launch": int(built_bytes), "achieved": round(built_bytes / sec / 1e9, 2) if sec > 0 else 0.0,
                        "frac": round(built_bytes / sec / 1e9 / HBM_PEAK_GBS, 4) if sec > 0 else 0.0,
                        "what": "14 B/record in + 8 B/key out + 32 B x rows + index"},
           "stage_ms_per_step": {k: round(v / subs, 4) for k, v in zip(("stream", None, "scatter", "hist"), st["stage_ms"]) if k},
           "resident_hot_path_M_alignments_per_s": round(n_records * steps / wall / 1e6, 1),
           "hits_fraction": round(int(res["cnt"][9]) / max(int(res["cnt"][0]), 1), 4)}
    traffic = None
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj):
        try:
            tr = json.load(open(tj))
            per_rec = tr.get("k_stream_bytes_per_record")
            traffic = int(per_rec * n_records) if per_rec else tr.get("k_stream_bytes_per_launch")
        except Exception:
            traffic = None
    out["traffic"] = traffic
    out["traffic_source"] = ("not measured in this run: bytes per record from profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in passes of their own "
                             "over this same command, tools/profile_bench.sh; 2 x FETCH_SIZE + WRITE_SIZE per the gfx950 correction) x the records of a launch")
    # attainable bandwidth on THIS card: a plain 1 GiB device copy, best of 5
    src = torch.empty(1 << 28, dtype=torch.int32, device=dev).fill_(1)
    dst = torch.empty_like(src)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ms = []
    for _ in range(6):
        ev[0].record()
        dst.copy_(src)
        ev[1].record()
        torch.cuda.synchronize()
        ms.append(ev[0].elapsed_time(ev[1]))
    copy_gbs = 2 * src.numel() * 4 / (min(ms[1:]) * 1e-3) / 1e9
    del src, dst
    out["copy_kernel_GBps"] = round(copy_gbs, 1)
    out["frac_of_copy_kernel"] = round(out["achieved"] / copy_gbs, 4)
    checks = {"cnt0_equals_records": bool(int(res["cnt"][0]) == n_records * steps),
              "rep_sum_equals_cnt9": bool(int(res["rep_cnt"][: len(rep_len)].sum()) == int(res["cnt"][9]))}
    if with_check:
        # a prefix of the same records through the oracle (the checker): the GPU path must give its numbers exactly
        from oracle import binding as orc
        m = min(4_000_000, n_records)
        h = {k: v[:m].cpu().numpy() for k, v in d.items()}
        ot = orc.OracleTable(cs, rep_len, len(tb.fams), len(tb.clas))
        ot.add_rows(tb.chrom, tb.start, tb.end, tb.cons_start, tb.cons_end, tb.rep_name, tb.fam_of_row, tb.cla_of_row)
        flag16 = np.where(h["flag5"] & 8, 16, 0).astype(np.uint16)
        want = ot.run({}, list(range(len(chroms))), h["tid"], h["pos"], h["tmpend"], h["mapq"], flag16, want_hits=False)
        e.reset()
        e.submit_device(ptrs, m, stream=stream)
        got = e.finish()
        checks["resident_sample_matches_oracle"] = bool(all(np.array_equal(got[k], want[k]) for k in ("cnt", "rep_cnt", "fam_cnt", "cla_cnt", "cov", "cov_uniq")))
        # "a fair multi-core CPU number for our restatement": the oracle (the hot path only: decoded records in,
        # counters and coverage out) on every core of this process's CPU share — contiguous shards, private accumulators, the
        # read-only table shared; something the reference's process-global state cannot do
        nthr = max(1, min(len(os.sched_getaffinity(0)), 16))
        from concurrent.futures import ThreadPoolExecutor
        mm = min(n_records, m * 4)
        hh = {k: v[:mm].cpu().numpy() for k, v in d.items()}
        fl_all = np.where(hh["flag5"] & 8, 16, 0).astype(np.uint16)
        cuts = [(mm * t // nthr, mm * (t + 1) // nthr) for t in range(nthr)]

        def one(b):
            lo, hi = b
            return ot.run({}, list(range(len(chroms))), hh["tid"][lo:hi], hh["pos"][lo:hi], hh["tmpend"][lo:hi], hh["mapq"][lo:hi], fl_all[lo:hi], want_hits=False)["cnt"]
        t2 = time.perf_counter()
        one((0, m))
        st_s = time.perf_counter() - t2
        t3 = time.perf_counter()
        with ThreadPoolExecutor(nthr) as ex:
            parts = list(ex.map(one, cuts))
        mt_s = time.perf_counter() - t3
        checks["cpu_mt_counts_all_records"] = bool(int(sum(int(c[0]) for c in parts)) == mm)
        out["cpu_baseline_mt"] = {"value": round(mm / mt_s / 1e6, 3), "unit": "M alignments/s", "cores": nthr, "kind": "port",
                                  "sample": (f"oracle/liboracle.so (our single-threaded C restatement of generic.c:745-1036) on the first {mm} DECODED records of the resident set in {nthr} "
                                             f"contiguous shards, one thread each, private accumulators, merge not timed ({mt_s:.1f} s); hot path only — no BAM decode, no files"),
                                  "one_thread_M_alignments_per_s": round(m / st_s / 1e6, 3)}
        ot.close()
    e.close()
    # configs[4]: the same kernel in filter mode (k_stream<ATOMIC_LOCUS>: derive + classify + one atomic per run of equal rows)
    if getattr(a, "filter_steps", 0) > 0:
        ef = eng.Engine(table, {"filter_mode": True}, batch_capacity=n_records)
        ef.set_tidmap(list(range(len(chroms))))
        ef.submit_device(ptrs, n_records, stream=stream)
        ef.sync()
        ef.reset()
        torch.cuda.synchronize()
        for _ in range(steps):
            ef.submit_device(ptrs, n_records, stream=stream)
        ef.sync()
        sf = ef.stats()
        rf = ef.finish()
        fms = sf["kernel_ms"] / max(sf["submits"], 1)                       # filter mode: one kernel per submit, timed by the engine's HIP events around it
        out["filter_kernel"] = {"kernel": "k_stream<ATOMIC_LOCUS> (derive + classify + per-locus count)", "avg_launch_ms": round(fms, 4), "launches": int(sf["submits"]),
                                "records_per_launch": n_records, "algorithmic_bytes_per_launch": int(survey_bytes),
                                "achieved": round(survey_bytes / (fms * 1e-3) / 1e9, 2) if fms > 0 else 0.0, "unit": "GB/s",
                                "frac": round(survey_bytes / (fms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if fms > 0 else 0.0,
                                "bytes_accounting": "SURVEY.md 8(d) K1 as for stat: 18 B/read (14 in + the 4-byte per-locus count) + 28 B x table rows + index; full table, every class"}
        checks["filter_locus_sum_equals_cnt9"] = bool(int(rf["locus_cnt"].astype(np.uint64).sum()) == int(rf["cnt"][9]))
        ef.close()
    table.close()
    return out, checks


# ------------------------------------------------------------------------------------------------ launcher
def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child process (nothing here has touched the GPU)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if rank == 0:
            log(f"--gpus {a.gpus} but WORLD_SIZE={world}: running with {world} ranks")
        a.gpus = world
    threads = a.threads or max(1, min(16, len(os.sched_getaffinity(0))))

    import torch
    import torch.distributed as dist
    from __graft_entry__ import build
    if rank == 0:
        build()
    # rehearsal hook for a one-GPU box (never set by the driver): every rank on cuda:0, gloo for the barriers, and the
    # command's ranks exchange their partials through files instead of RCCL (two RCCL ranks cannot share a device)
    share_gpu = os.environ.get("ITX_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist.barrier()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---------------------------------------------------------------- device memory settled before anything is timed
    settle = vram_settle(local_rank) if a.settle and os.environ.get("ITX_BENCH_SHARE_GPU") != "1" else None      # (ranks sharing one card would fight over it)

    # ---------------------------------------------------------------- inputs: rank 0 makes them, everybody learns where
    box = [None, None]
    if rank == 0:
        t0 = time.time()
        wd, info = ensure_inputs(a, threads)
        info["setup_s"] = round(time.time() - t0, 1)
        box = [wd, info]
        log(f"inputs in {wd}: {info}")
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    wd, info = box
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), ITX_TIMING="1", ITX_GPUS="1")      # --gpus decides, not the node: the ranks are this launcher's
    env.pop("ITX_RANK", None)
    scratch = os.path.join(wd, f"run_rank{rank}")
    shutil.rmtree(scratch, ignore_errors=True)
    os.makedirs(scratch, exist_ok=True)

    # the file list of the weak-scaling run: N hard links to the one BAM (same bytes, N x the reads)
    bam = os.path.join(wd, "reads.bam")
    if world > 1 and rank == 0:
        for r in range(world):
            ln = os.path.join(wd, f"reads_copy{r}.bam")
            if not os.path.exists(ln):
                os.link(bam, ln)
    if world > 1:
        dist.barrier()
    weak_list = ",".join(os.path.join(wd, f"reads_copy{r}.bam") for r in range(world)) if world > 1 else bam

    # a name no earlier (crashed) invocation can have left behind: the ranks of a run find each other through this file
    nonce = [f"{os.getpid()}_{int(time.time())}"]
    if world > 1:
        dist.broadcast_object_list(nonce, src=0)

    def one_run(aln, tag, step):
        """one whole command over `aln` by all ranks; returns this rank's wall seconds and its stderr"""
        e = dict(env)
        if world > 1:
            e.update(ITX_RANK=str(rank), ITX_WORLD=str(world), ITX_DEVICE=str(local_rank), ITX_COMM_ID=os.path.join(wd, f"comm_{nonce[0]}_{tag}_{step}.id"),
                     ITX_EXCHANGE="file" if share_gpu else "rccl", ITX_COMM_TIMEOUT=os.environ.get("ITX_COMM_TIMEOUT", "300"))
        wall, rc, err, seen = run_timed(OURS, base_args(wd) + [aln], scratch, e, (SCAN_BEGIN, SCAN_END))
        if rc != 0:
            raise RuntimeError(f"rank {rank}: iteres stat failed ({rc}): {err[-800:]}")
        return wall, err, seen

    # ---------------------------------------------------------------- warmup, then EXACTLY K timed steps
    for w in range(a.warmup):
        one_run(weak_list, "warm", w)
    fence()
    t1 = time.perf_counter()
    last = None
    step_walls = []
    slowest = None
    for k in range(a.steps):
        last = one_run(weak_list, "step", k)
        step_walls.append(round(last[0], 3))
        if slowest is None or last[0] > slowest[0]:
            slowest = last
        if world > 1:
            dist.barrier()
    fence()
    elapsed = time.perf_counter() - t1
    rep = report_counts(os.path.join(scratch, "out.iteres.report")) if rank == 0 else []
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if not share_gpu else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    total_reads = a.reads * world * a.steps

    # strong scaling beside it: the ONE BAM split over the ranks (N > 1 only; two runs, the second timed)
    strong = None
    if world > 1:
        one_run(bam, "strongwarm", 0)
        fence()
        ts = time.perf_counter()
        one_run(bam, "strong", 0)
        fence()
        dt = time.perf_counter() - ts
        tm = torch.tensor([dt], dtype=torch.float64, device=dev if not share_gpu else "cpu")
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        strong = {"what": f"the one {a.reads}-read BAM split over {world} ranks (configs[3]), whole command", "wall_s": round(float(tm.item()), 3),
                  "M_alignments_per_s": round(a.reads / float(tm.item()) / 1e6, 2)}

    out = None
    if rank == 0:
        wall_last, err_last, seen = last
        scan_s = seen.get(SCAN_END, 0) - seen.get(SCAN_BEGIN, 0) if SCAN_BEGIN in seen and SCAN_END in seen else None
        phases = [ln for ln in err_last.split("\n") if ln.startswith("[itx timing]")]
        # "total reads (pair)" (generic.c:53-70) counts first ends: every record of a single-end file; with --paired 1 half of them
        # plus the odd read a chromosome's segment may end on (tools/mkbam.c), a few thousand at most
        checks = {"report_total_equals_reads": bool(rep and (rep[0] == a.reads * world if not a.paired
                                                             else 0 <= rep[0] - (a.reads * world) // 2 <= 4096 * world)),
                  "outputs_written": all(os.path.exists(os.path.join(scratch, fn)) for fn in TEXT_OUTPUTS + ("out.iteres.bigWig", "out.iteres.unique.bigWig"))}
        ms_per_step = elapsed * 1e3 / a.steps
        out = {
            "metric": "M alignments/sec through `iteres stat` (hg38 rmsk) at 1/2/4/8 MI355X vs CPU ref",
            "value": round(total_reads / elapsed / 1e6, 3), "unit": "M alignments/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[2]: `iteres stat -w` end to end on a {a.reads}-read coordinate-sorted synthetic hg38 BAM "
                                    f"({a.seq_len} bases + qualities per record, content={a.content}, cigar={a.cigar}{', paired' if a.paired else ''}, {info['bam_bytes'] / 1e9:.1f} GB BGZF) vs {a.rows}-row rmsk, reference defaults, "
                                    "per-base coverage wigs on; one step = one whole run of the command")
                       if world == 1 else
                       (f"BASELINE configs[3], weak: `iteres stat -w` on a list of {world} such {a.reads}-read BAMs over {world} GPUs (one process per GPU, "
                        "shares of the compressed bytes, one RCCL sum-reduce of the partial onto rank 0, rank 0 writes the files)"),
                       "content": dict({"mode": a.content, "cigar": a.cigar, "paired": int(a.paired), "pileup_loci": int(a.pileup),
                                         "what": {"hiseq": "independent bases per nibble, 40-value HiSeq-like qualities, Illumina-style names: literal-heavy DEFLATE streams — the content `value` is timed on",
                                                  "novaseq": "independent bases, 4-bin run-structured qualities",
                                                  "legacy": "round 2's file: one base per byte pair and a period-32 quality pattern — 97 % of the bytes come out of LZ77 matches (a best case)"}[a.content]},
                                        **(info.get("content") or {})),
                       "reads_per_step": a.reads * world, "seq_len": a.seq_len, "bam_bytes": info["bam_bytes"], "rmsk_rows": a.rows,
                       "host_threads_per_rank": threads, "host_cores_visible": os.cpu_count(),
                       "timed_region": "K runs of the whole command (process start to exit), files in the page cache"},
            "scan_only": {"what": "banner to banner (stat.c:144,153): device table build + BAM decode + classify + accumulate, last step",
                          "seconds": round(scan_s, 3) if scan_s else None,
                          "M_alignments_per_s": round(a.reads * world / scan_s / 1e6, 2) if scan_s else None},
            "step_wall_s": {"each": step_walls, "median": sorted(step_walls)[len(step_walls) // 2], "min": min(step_walls),
                            "note": "rank 0's wall of every timed step; `value` is the mean over all of them (boxes of the pool differ: on some a run loses a second or more in device allocations)"},
            "vram_settle": settle,
            "phases_last_step": phases,
            "phases_slowest_step": [ln for ln in slowest[1].split("\n") if ln.startswith("[itx timing]")] if slowest is not last else "the last step",
            "checks": checks,
            "inputs": info,
        }
        if strong:
            out["strong_scaling"] = strong

    # ---------------------------------------------------------------- CPU baseline: the reference binary on a bounded sample (rank 0, N = 1)
    if rank == 0 and world == 1 and a.cpu_reads > 0 and os.path.exists(REF):
        sample = os.path.join(wd, "sample.bam")
        renv = dict(os.environ)
        runs = []
        for k in range(max(1, a.cpu_repeats)):
            wall_r, rc_r, err_r, seen_r = run_timed(REF, base_args(wd) + [sample], os.path.join(wd, "ref_run"), renv, (SCAN_BEGIN, SCAN_END))
            scan_k = seen_r.get(SCAN_END, 0) - seen_r.get(SCAN_BEGIN, 0) if SCAN_BEGIN in seen_r and SCAN_END in seen_r else None
            runs.append((wall_r, scan_k, rc_r))
            log(f"reference run {k + 1}/{a.cpu_repeats}: {wall_r:.1f} s (rc {rc_r})")
        runs.sort(key=lambda x: x[0])
        wall_r, scan_r, rc_r = runs[len(runs) // 2]                            # the median run
        wall_o, rc_o, err_o, _ = run_timed(OURS, base_args(wd) + [sample], os.path.join(wd, "ours_sample"), env)
        same = {fn: (os.path.exists(os.path.join(wd, "ours_sample", fn)) and os.path.exists(os.path.join(wd, "ref_run", fn))
                     and filecmp.cmp(os.path.join(wd, "ref_run", fn), os.path.join(wd, "ours_sample", fn), shallow=False)) for fn in TEXT_OUTPUTS}
        fixed_r = wall_r - scan_r if scan_r else None
        full_est = (fixed_r + a.reads / (a.cpu_reads / scan_r)) if scan_r else None
        out["cpu_baseline"] = {
            "value": round(a.cpu_reads / wall_r / 1e6, 4), "unit": "M alignments/s", "cores": 1, "kind": "reference",
            "sample": (f"oracle/_ref/iteres stat -w (the reference's own sources, -O as its makefile, single-threaded like the reference) on a {a.cpu_reads}-read BAM "
                       f"from the same generator / table; median of {len(runs)} runs ({wall_r:.1f} s whole command, rc {rc_r})"),
            "runs_wall_s": [round(r[0], 2) for r in runs],
            "scan_only_M_alignments_per_s": round(a.cpu_reads / scan_r / 1e6, 4) if scan_r else None,
            "fixed_s": round(fixed_r, 2) if fixed_r else None,
            "extrapolated_full_size": {"what": f"fixed_s + {a.reads} reads / scan rate (linear, BASELINE.md 3.3) — arithmetic, not a measurement",
                                       "seconds": round(full_est, 1) if full_est else None,
                                       "M_alignments_per_s": round(a.reads / full_est / 1e6, 4) if full_est else None},
            "drop_in_same_sample_wall_s": round(wall_o, 2),
            "host_cores_visible": os.cpu_count(),
        }
        # the same sources at -O2 (reported separately), one run
        if os.path.exists(REF_O2):
            wall_2, rc_2, _, seen_2 = run_timed(REF_O2, base_args(wd) + [sample], os.path.join(wd, "ref_run_o2"), renv, (SCAN_BEGIN, SCAN_END))
            scan_2 = seen_2.get(SCAN_END, 0) - seen_2.get(SCAN_BEGIN, 0) if SCAN_BEGIN in seen_2 and SCAN_END in seen_2 else None
            out["cpu_baseline"]["O2_build"] = {"what": "oracle/_ref/o2/iteres: the same sources with -O2 instead of the makefile's -O, one run on the same sample",
                                               "wall_s": round(wall_2, 2), "M_alignments_per_s": round(a.cpu_reads / wall_2 / 1e6, 4), "rc": rc_2,
                                               "scan_only_M_alignments_per_s": round(a.cpu_reads / scan_2 / 1e6, 4) if scan_2 else None}
        out["files_identical"] = same
        out["checks"]["sample_outputs_identical_to_reference"] = bool(rc_r == 0 and rc_o == 0 and all(same.values()))
        if full_est:
            out["speedup_vs_cpu_reference"] = {"same_sample_measured": round(wall_r / wall_o, 1),
                                               "whole_command_at_full_size_extrapolated": round(out["value"] / (a.reads / full_est / 1e6), 1),
                                               "note": "only the same-sample figure is a measurement of both programs; the other divides by the extrapolation above"}
    elif rank == 0 and world == 1:
        out["cpu_baseline"] = None

    # ---------------------------------------------------------------- configs[4]: `iteres filter -c <class>` on the same BAM (rank 0, N = 1)
    if rank == 0 and world == 1 and a.filter_steps > 0 and info.get("filter_class"):
        fc = info["filter_class"]
        fargs = lambda aln: ["filter", "-c", fc["name"], "-o", "out", os.path.join(wd, "chrom.sizes"), os.path.join(wd, "rep.sizes"), os.path.join(wd, "rmsk.txt"), aln]
        fdir = os.path.join(wd, "filter_run")
        walls, ferr = [], ""
        for k in range(a.filter_steps + 1):                                   # the first run is the warm-up
            w, rc_f, ferr, _ = run_timed(OURS, fargs(bam), fdir, env)
            if rc_f != 0:
                raise RuntimeError(f"iteres filter failed ({rc_f}): {ferr[-800:]}")
            if k:
                walls.append(round(w, 3))
        leg = {"what": f"`iteres filter -c {fc['name']}` (the biggest class: {fc['rows']} rows = {fc['fraction_of_rows']:.1%} of the table) end to end on the same {a.reads}-read BAM",
               "wall_s": walls, "M_alignments_per_s": round(a.reads / (sum(walls) / len(walls)) / 1e6, 2),
               "phases_last_run": [ln for ln in ferr.replace("\r", "\n").split("\n") if ln.startswith("[itx timing]")]}
        if a.cpu_reads > 0 and os.path.exists(REF):
            sample = os.path.join(wd, "sample.bam")
            w_r, rc_r, _, _ = run_timed(REF, fargs(sample), os.path.join(wd, "filter_ref"), dict(os.environ))
            w_o, rc_o, _, _ = run_timed(OURS, fargs(sample), os.path.join(wd, "filter_ours_sample"), env)
            names = [f"out_{fc['name']}.iteres.loci", f"out_{fc['name']}.iteres.reportloci"]
            fsame = {fn: (os.path.exists(os.path.join(wd, "filter_ref", fn)) and os.path.exists(os.path.join(wd, "filter_ours_sample", fn))
                          and filecmp.cmp(os.path.join(wd, "filter_ref", fn), os.path.join(wd, "filter_ours_sample", fn), shallow=False)) for fn in names}
            leg["reference_on_sample"] = {"wall_s": round(w_r, 2), "M_alignments_per_s": round(a.cpu_reads / w_r / 1e6, 4), "rc": rc_r,
                                          "drop_in_same_sample_wall_s": round(w_o, 2), "files_identical": fsame}
            out["checks"]["filter_sample_outputs_identical_to_reference"] = bool(rc_r == 0 and rc_o == 0 and all(fsame.values()))
        out["filter_leg"] = leg

    # ---------------------------------------------------------------- roofline of the overlap kernel, records resident in HBM (rank 0)
    if rank == 0 and a.replay_steps > 0:
        n_res = a.replay_reads or a.reads
        roof, rchecks = resident_roofline(a, n_res, a.replay_steps, rank, with_check=(world == 1 and not a.no_replay_check))
        for k in ("cpu_baseline_mt", "filter_kernel"):                        # measured inside the replay, reported beside it
            if k in roof:
                (out["filter_leg"] if k == "filter_kernel" and "filter_leg" in out else out)[k] = roof.pop(k)
        out["roofline"] = roof
        out["checks"].update(rchecks)
    if rank == 0:
        print(json.dumps(out), flush=True)
        if not a.keep:
            shutil.rmtree(scratch, ignore_errors=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
