"""BASELINE.json's full-size configurations under `pytest -m gpu` (the small parity cases live in test_gpu_parity.py):

  configs[1]  50 M reads x 5.5 M-row table: engine identities, both accumulate paths, a 12 M-record prefix against the oracle
              record by record; the COMMAND on a 50 M-read BAM against the reference binary, every text output byte for byte
              (`stat -w` and `filter -n`).
  configs[2]  500 M reads, coverage on: size-independent properties — identities between the counters, the result does not
              depend on how the stream is cut into batches, two engines that share the stream and add their exported
              partials give the single engine's result (the N > 1 identity of configs[3], on one device); the command on a
              500 M-read BAM gives the same files whatever the chunking of the decoder.
  configs[4]  filter mode at 500 M: the same properties for the per-locus counts, and the command's .loci files.
  plus        a direct-vs-oracle case that forces far more than 65 535 keys into one partition (k_hist's split items).

Inputs are seeded and made on the box (iteres_amd/synth.py, tools/mkbam.c); nothing is read from /root/reference. The
reference binary (oracle/_ref/iteres, built in the container and shipped as a file) and the oracle are the checkers."""
import filecmp
import os
import subprocess

import numpy as np
import pytest

import enginecase as ec
from iteres_amd import engine as eng, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = os.path.join(ROOT, "iteres_amd", "host", "iteres")
REF = os.path.join(ROOT, "oracle", "_ref", "iteres")
MKBAM = os.path.join(ROOT, "tools", "mkbam")
N_ROWS = 5_500_000
THREADS = str(max(1, min(16, len(os.sched_getaffinity(0)))))


@pytest.fixture(scope="module")
def big():
    """the hg38-scale table of bench.py: 5.5 M rows, 15 k names, 60 families, 20 classes"""
    chroms = synth.HG38_CHROMS
    tb = synth.make_table(20260101, chroms, N_ROWS, n_names=15000, n_fams=60, n_clas=20, overlap_frac=0.02)
    rep_len = np.array([tb.rep_len.get(n, 0) for n in tb.names], np.uint32)
    rows = eng.make_rows(tb.chrom, tb.start, tb.end, tb.cons_start, tb.cons_end, tb.rep_name, tb.fam_of_row, tb.cla_of_row)
    cs = np.array([s for _, s in chroms], np.int64)
    table = eng.Table(rows, cs, rep_len, len(tb.fams), len(tb.clas))
    yield dict(chroms=chroms, tb=tb, rep_len=rep_len, rows=rows, cs=cs, table=table, nf=len(tb.fams), nc=len(tb.clas), t2c=list(range(len(chroms))))
    table.close()


@pytest.fixture(scope="module")
def files(big, tmp_path_factory):
    """the table as files + BAMs from tools/mkbam.c: the 50 M-read one (both programs read it) carries 50 bases + 40-value qualities
    per record, independent bases, Illumina-style names and 5 % CIGARs with S / D / I / N — what a BAM looks like to a DEFLATE
    decoder and to bam_calend; the 500 M-read one (chunking of the command alone) carries none"""
    wd = str(tmp_path_factory.mktemp("scale"))
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-o", MKBAM, os.path.join(ROOT, "tools", "mkbam.c"), "-lz", "-ldl"])
    synth.write_sizes(os.path.join(wd, "chrom.sizes"), big["chroms"])
    synth.write_sizes(os.path.join(wd, "rep.sizes"), big["tb"].rep_len.items())
    synth.write_rmsk(os.path.join(wd, "rmsk.txt"), big["tb"], workers=int(THREADS))
    env = dict(os.environ, OMP_NUM_THREADS=THREADS)
    subprocess.check_call([MKBAM, os.path.join(wd, "chrom.sizes"), "50000000", os.path.join(wd, "r50M.bam"), "50", "11", "0", "content=hiseq", "cigar=mixed"], env=env)
    subprocess.check_call([MKBAM, os.path.join(wd, "chrom.sizes"), "500000000", os.path.join(wd, "r500M.bam"), "0", "11"], env=env)
    return wd


def _ptrs(d, lo=0):
    return {k: v.data_ptr() + lo * v.element_size() for k, v in d.items()}


def _identities(res, n_records, n_rep, nf, nc):
    c = res["cnt"]
    assert int(c[0]) == n_records and int(c[1]) == 0                 # single-end: every record is an "end 1"
    assert int(c[2]) == int(c[4]) == int(c[6]) == n_records          # all mapped, all on known chromosomes
    assert int(res["rep_cnt"][:n_rep].sum()) == int(c[9]) == int(res["fam_cnt"][:nf].sum()) == int(res["cla_cnt"][:nc].sum())
    assert int(res["rep_cnt"][n_rep:].sum()) == int(c[10]) == int(res["fam_cnt"][nf:].sum()) == int(res["cla_cnt"][nc:].sum())
    assert int(c[10]) <= int(c[7]) <= int(c[6]) and int(c[9]) <= int(c[6])


def _run_device(table, params, d, n, pieces, cap=None):
    """records [0, n) of the device arrays `d` through one engine, cut into `pieces` submits"""
    import torch
    e = eng.Engine(table, params, batch_capacity=cap or -(-n // pieces) + 16)
    e.set_tidmap(list(range(len(synth.HG38_CHROMS))))
    step = -(-n // pieces)
    step = -(-step // 16) * 16                                       # device batches start on 16-record boundaries
    st = torch.cuda.current_stream().cuda_stream
    for lo in range(0, n, step):
        e.submit_device(_ptrs(d, lo), min(step, n - lo), stream=st)
    res = e.finish()
    e.close()
    return res


def test_config1_engine_50M(big):
    """configs[1]: 50 M coordinate-sorted reads vs the 5.5 M-row table — counter identities, partition path == atomic path,
    and the first 12 M records equal to the oracle, chosen row by chosen row."""
    import torch
    from oracle import binding as orc
    n, m = 50_000_000, 12_000_000
    dev = torch.device("cuda", 0)
    d = synth.make_reads_device(20260102, big["chroms"], n, dev)
    n_rep = len(big["rep_len"])
    res_p = _run_device(big["table"], dict(accum=eng.ACCUM_PARTITION), d, n, 1)
    res_a = _run_device(big["table"], dict(accum=eng.ACCUM_ATOMIC), d, n, 1)
    for k in ("cnt", "rep_cnt", "fam_cnt", "cla_cnt", "cov", "cov_uniq"):
        assert np.array_equal(res_p[k], res_a[k]), k
    _identities(res_p, n, n_rep, big["nf"], big["nc"])
    assert 0.5 < int(res_p["cnt"][9]) / n < 0.9                       # ~73 % of the reads land in repeats
    # the prefix against the oracle, per record
    h = {k: v[:m].cpu().numpy() for k, v in d.items()}
    ot = ec.oracle_table(big["rows"], big["cs"], big["rep_len"], big["nf"], big["nc"])
    want = ot.run({}, big["t2c"], h["tid"], h["pos"], h["tmpend"], h["mapq"], np.where(h["flag5"] & 8, 16, 0).astype(np.uint16), want_hits=True)
    ot.close()
    hits = torch.empty(m, dtype=torch.int32, device=dev)
    e = eng.Engine(big["table"], {}, batch_capacity=m)
    e.set_tidmap(big["t2c"])
    e.submit_device(_ptrs(d), m, hit_ptr=hits.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    got = e.finish()
    e.close()
    assert np.array_equal(hits.cpu().numpy().astype(np.int64), want["hit_row"]), "chosen rows differ from the oracle"
    for k in ("cnt", "rep_cnt", "fam_cnt", "cla_cnt", "cov", "cov_uniq"):
        assert np.array_equal(got[k], want[k]), k


def _cli(exe, head, wd, bam, out_dir, env=None):
    os.makedirs(out_dir, exist_ok=True)
    args = [exe] + head + ["-o", "out", os.path.join(wd, "chrom.sizes"), os.path.join(wd, "rep.sizes"), os.path.join(wd, "rmsk.txt"), os.path.join(wd, bam)]
    pr = subprocess.run(args, cwd=out_dir, capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS=THREADS, **(env or {})))
    assert pr.returncode == 0, pr.stderr[-2000:]
    return pr


def _same_files(a, b, skip_suffix=(".bigWig",)):
    names = sorted(fn for fn in os.listdir(a) if not fn.endswith(skip_suffix))
    assert names and names == sorted(fn for fn in os.listdir(b) if not fn.endswith(skip_suffix)), (os.listdir(a), os.listdir(b))
    for fn in names:
        assert filecmp.cmp(os.path.join(a, fn), os.path.join(b, fn), shallow=False), fn
    return names


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/iteres (the reference, built in the container) did not travel")
def test_config1_cli_50M_vs_reference(files):
    """configs[1] through the command, next to the reference binary on the same 50 M-read BAM: .subfamily/.family/.class
    stat, .report, both per-base wigs (stat -w) and .loci/.reportloci (filter -n) byte for byte."""
    _cli(REF, ["stat", "-w"], files, "r50M.bam", os.path.join(files, "ref_stat"))
    _cli(OURS, ["stat", "-w"], files, "r50M.bam", os.path.join(files, "our_stat"))
    names = _same_files(os.path.join(files, "ref_stat"), os.path.join(files, "our_stat"))
    assert len(names) == 6
    for fn in ("out.iteres.bigWig", "out.iteres.unique.bigWig"):
        assert os.path.getsize(os.path.join(files, "our_stat", fn)) > 0
    _cli(REF, ["filter", "-n", "Rep1"], files, "r50M.bam", os.path.join(files, "ref_filter"))
    _cli(OURS, ["filter", "-n", "Rep1"], files, "r50M.bam", os.path.join(files, "our_filter"))
    assert len(_same_files(os.path.join(files, "ref_filter"), os.path.join(files, "our_filter"))) == 2


def _add_partials(engines, dev):
    """what a multi-GPU driver does with the exported partials, here with torch on one device: sum them"""
    import torch
    n64, n32 = engines[0].partial_size()
    s64 = torch.zeros(n64, dtype=torch.int64, device=dev)
    s32 = torch.zeros(n32, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for e in engines:
        assert e.partial_size() == (n64, n32)
        p64 = torch.zeros(n64, dtype=torch.int64, device=dev)
        p32 = torch.zeros(n32, dtype=torch.int32, device=dev)
        e.export_partial(p64.data_ptr(), p32.data_ptr(), stream=st)
        e.sync()
        torch.cuda.synchronize()
        s64 += p64
        s32 += p32                                                   # two's-complement adds: the sums mod 2^64 / 2^32
    torch.cuda.synchronize()
    return s64, s32


@pytest.mark.parametrize("filter_mode", [False, True])
def test_config2_and_4_properties_500M(big, filter_mode):
    """configs[2] (stat, coverage on) and configs[4] (filter) at 500 M records resident in HBM: identities, batch-split
    invariance, and the N > 1 identity — two engines take 50 M / 450 M of the stream, their exported partials are added,
    finish_partial gives the single engine's result."""
    import torch
    n, cut = 500_000_000, 50_000_000
    dev = torch.device("cuda", 0)
    d = synth.make_reads_device(20260103, big["chroms"], n, dev)
    n_rep = len(big["rep_len"])
    params = dict(filter_mode=filter_mode)
    keys = ("cnt", "locus_cnt") if filter_mode else ("cnt", "rep_cnt", "fam_cnt", "cla_cnt", "cov", "cov_uniq")
    one = _run_device(big["table"], params, d, n, 1)
    ten = _run_device(big["table"], params, d, n, 10)
    for k in keys:
        assert np.array_equal(one[k], ten[k]), k
    c = one["cnt"]
    assert int(c[0]) == n and int(c[6]) == n
    if filter_mode:
        assert int(one["locus_cnt"][: N_ROWS].astype(np.uint64).sum()) == int(c[9])
    else:
        _identities(one, n, n_rep, big["nf"], big["nc"])
        # every read in a repeat of known consensus length that starts inside the repeat adds >= 1 base of coverage
        assert int(one["cov"].astype(np.uint64).sum()) > int(c[9])
        assert int(one["cov_uniq"].astype(np.uint64).sum()) <= int(one["cov"].astype(np.uint64).sum())
    del ten
    # two engines, one stream
    st = torch.cuda.current_stream().cuda_stream
    ea = eng.Engine(big["table"], params, batch_capacity=cut)
    eb = eng.Engine(big["table"], params, batch_capacity=n - cut)
    for e in (ea, eb):
        e.set_tidmap(big["t2c"])
    ea.submit_device(_ptrs(d), cut, stream=st)
    eb.submit_device(_ptrs(d, cut), n - cut, stream=st)
    s64, s32 = _add_partials([ea, eb], dev)
    both = ea.finish_partial(s64.data_ptr(), s32.data_ptr())
    for k in keys:
        assert np.array_equal(both[k], one[k]), k
    ea.close()
    eb.close()


def test_config2_and_4_cli_500M_chunking(files):
    """configs[2] / configs[4] through the command on a 500 M-read BAM: the files must not depend on how the decoder cuts
    the stream (chunks of 128 MiB by default, 48 MiB and windows of 5000 blocks here), and the report must account for
    every read."""
    alt = dict(ITX_BGZF_CHUNK=str(48 << 20), ITX_DEV_WINDOW_BLOCKS="5000")
    _cli(OURS, ["stat", "-w"], files, "r500M.bam", os.path.join(files, "s500_a"))
    _cli(OURS, ["stat", "-w"], files, "r500M.bam", os.path.join(files, "s500_b"), alt)
    names = _same_files(os.path.join(files, "s500_a"), os.path.join(files, "s500_b"), skip_suffix=())
    assert len(names) == 8                                            # bigWigs included: same program, same bytes
    first = open(os.path.join(files, "s500_a", "out.iteres.report")).readline()
    assert first.strip().endswith(": 500000000"), first
    _cli(OURS, ["filter", "-n", "Rep1"], files, "r500M.bam", os.path.join(files, "f500_a"))
    _cli(OURS, ["filter", "-n", "Rep1"], files, "r500M.bam", os.path.join(files, "f500_b"), alt)
    assert len(_same_files(os.path.join(files, "f500_a"), os.path.join(files, "f500_b"))) == 2
    # the 50 M-read file is a different draw, so only the shape is comparable: same rows listed, counts ~10x
    assert os.path.getsize(os.path.join(files, "f500_a", "out_Rep1.iteres.loci")) > 0


@pytest.mark.parametrize("filter_mode", [False, True])
def test_two_engines_partials_vs_oracle(filter_mode):
    """The N > 1 identity pinned to the oracle at test size: two engines on one device take the halves of a stream, their
    exported partials are added, itx_engine_finish_partial must equal the single engine AND the oracle (stat and filter)."""
    import torch
    chroms = [("c1", 40_000_000), ("c2", 25_000_000)]
    t = synth.make_table(71, chroms, 60_000, n_names=400, n_fams=30, n_clas=10, overlap_frac=0.05, shuffle_frac=0.02)
    rl = np.array([t.rep_len.get(n, 0) for n in t.names], np.uint32)
    rows = eng.make_rows(t.chrom, t.start, t.end, t.cons_start, t.cons_end, t.rep_name, t.fam_of_row, t.cla_of_row)
    cs = np.array([s for _, s in chroms], np.int64)
    tid, pos, tmpend, mapq, f5 = synth.make_reads_soa(72, chroms, 600_000)
    params = dict(filter_mode=filter_mode)
    ot = ec.oracle_table(rows, cs, rl, len(t.fams), len(t.clas))
    want = ot.run(params, [0, 1], tid, pos, tmpend, mapq, np.where(f5 & 8, 16, 0).astype(np.uint16), want_hits=False)
    ot.close()
    tab = eng.Table(rows, cs, rl, len(t.fams), len(t.clas))
    half = 300_016
    engines = []
    for lo, hi in ((0, half), (half, len(tid))):
        e = eng.Engine(tab, params, batch_capacity=1 << 17)
        e.set_tidmap([0, 1])
        e.submit_host(tid[lo:hi], pos[lo:hi], tmpend[lo:hi], mapq[lo:hi], f5[lo:hi])
        engines.append(e)
    single = eng.Engine(tab, params, batch_capacity=1 << 18)
    single.set_tidmap([0, 1])
    single.submit_host(tid, pos, tmpend, mapq, f5)
    one = single.finish()
    s64, s32 = _add_partials(engines, torch.device("cuda", 0))
    both = engines[1].finish_partial(s64.data_ptr(), s32.data_ptr())
    keys = ("cnt", "locus_cnt") if filter_mode else ("cnt", "rep_cnt", "fam_cnt", "cla_cnt", "cov", "cov_uniq")
    for k in keys:
        a, b, w = both[k], one[k], want[k]
        if k == "locus_cnt":
            a, b, w = a[: len(rows)], b[: len(rows)], w[: len(rows)]
        assert np.array_equal(a, b), k
        assert np.array_equal(a, w), k
    for e in engines + [single]:
        e.close()
    tab.close()


@pytest.mark.parametrize("log_w", [None, 15])
def test_hot_partition_split_items_vs_oracle(log_w, monkeypatch):
    """(log_w 15: the same through the wide-partition kernels — several LDS windows per partition, the slot space ending
    inside the first.) k_hist's split-item branch (a partition with more than 65 535 keys is cut into several items, which then add their
    differences with atomics instead of owning the partition): one short consensus, 300 k reads that all choose rows of it —
    every key of the batch lands in ONE partition — compared with the oracle directly, both accumulate paths."""
    if log_w:
        monkeypatch.setenv("ITX_PART_LOGW", str(log_w))
    chroms = [("c1", 30_000_000)]
    rng = np.random.default_rng(5)
    n_rows = 20_000
    start = np.sort(rng.integers(0, 29_000_000, n_rows)).astype(np.int64)
    start = np.unique(start // 1500) * 1500                           # disjoint rows of <= 900 bp
    n_rows = len(start)
    end = start + rng.integers(200, 900, n_rows)
    cons_start = rng.integers(0, 200, n_rows)
    cons_end = cons_start + (end - start)
    z = np.zeros(n_rows, np.int32)
    rows = eng.make_rows(z, start, end, cons_start, cons_end, z, z, z)          # ONE name / family / class
    rl = np.array([1000], np.uint32)                                            # 1001 consensus slots: one partition
    cs = np.array([30_000_000], np.int64)
    n = 300_000
    pick = rng.integers(0, n_rows, n)
    pos = (start[pick] + rng.integers(-60, 300, n)).clip(0).astype(np.int32)
    order = np.argsort(pos, kind="stable")
    pos = pos[order]
    rd = dict(tid=np.zeros(n, np.int32), pos=pos, tmpend=(pos + rng.integers(36, 151, n)).astype(np.int32),
              mapq=rng.choice(np.array([0, 3, 20, 60], np.uint8), n), flag=np.where(rng.random(n) < 0.5, 16, 0).astype(np.uint16),
              mpos=np.zeros(n, np.int32), isize=np.zeros(n, np.int32))
    for accum in (eng.ACCUM_PARTITION, eng.ACCUM_ATOMIC):
        eres, ores, hits = ec.run_both(rows, cs, rl, 1, 1, {}, [0], rd, batch_capacity=n, accum=accum)
        ec.assert_same(eres, ores, hits, False, n_rows)
        assert int(eres["cnt"][9]) > 4 * 65_535                                  # far more keys than one item holds


def _long_consensus_case(seed, n_names, rep_len, n_rows, n_reads, long_frac):
    """Rows of a few hundred bases and (long_frac of them) of 9-25 kb on one chromosome, consensus ranges as long as the rows;
    reads of 36-150 bases and fragments of 9-20 kb (coverage ranges of more than 8192 slots)."""
    rng = np.random.default_rng(seed)
    size = 120_000_000
    start = np.unique(rng.integers(0, size - 40_000, n_rows) // 30_000) * 30_000            # disjoint, 30 kb apart
    n_rows = len(start)
    is_long = rng.random(n_rows) < long_frac
    length = np.where(is_long, rng.integers(9_000, 25_000, n_rows), rng.integers(200, 900, n_rows))
    end = start + length
    rep = rng.integers(0, n_names, n_rows).astype(np.int32)
    cons_start = np.array([rng.integers(0, max(1, rep_len - int(l))) for l in length], np.int64)
    cons_end = cons_start + length
    z = np.zeros(n_rows, np.int32)
    rows = eng.make_rows(z, start, end, cons_start, cons_end, rep, rep % 7, rep % 3)
    pick = rng.integers(0, n_rows, n_reads)
    frag = rng.random(n_reads) < 0.15
    pos = (start[pick] + rng.integers(-100, 400, n_reads)).clip(0).astype(np.int32)
    ln = np.where(frag & is_long[pick], rng.integers(9_000, 20_000, n_reads), rng.integers(36, 151, n_reads))
    order = np.argsort(pos, kind="stable")
    pos, ln = pos[order], ln[order]
    rd = dict(tid=np.zeros(n_reads, np.int32), pos=pos, tmpend=(pos + ln).astype(np.int32),
              mapq=rng.choice(np.array([0, 3, 20, 60], np.uint8), n_reads), flag=np.where(rng.random(n_reads) < 0.5, 16, 0).astype(np.uint16),
              mpos=np.zeros(n_reads, np.int32), isize=np.zeros(n_reads, np.int32))
    return rows, np.array([size], np.int64), np.full(n_names, rep_len, np.uint32), rd, n_rows


@pytest.mark.parametrize("log_w", [14, 15, 16])
def test_wide_partitions_vs_oracle(log_w, monkeypatch):
    """Partitions of 2^14 .. 2^16 slots (k_hist walks a partition's LDS windows one after the other; k_stream cuts a coverage
    range of 8192 slots or more into two keys even inside one partition) forced on a small slot space, against the oracle."""
    monkeypatch.setenv("ITX_PART_LOGW", str(log_w))
    rows, cs, rl, rd, n_rows = _long_consensus_case(40 + log_w, 40, 30_000, 3000, 200_000, 0.3)
    eres, ores, hits = ec.run_both(rows, cs, rl, 7, 3, {}, [0], rd, batch_capacity=len(rd["pos"]), accum=eng.ACCUM_PARTITION)
    ec.assert_same(eres, ores, hits, False, n_rows)
    assert int(eres["cnt"][9]) > 150_000


def test_slot_space_of_40M_takes_the_partition_path_vs_oracle():
    """5000 names x 8001 consensus slots = 40 M slots: more than 4096 windows of 8192. An explicit ITX_ACCUM_PARTITION request
    used to fail here (and the default fell back to device atomics); now the partitions are two windows wide. Against the
    oracle, and the default path gives the same."""
    rows, cs, rl, rd, n_rows = _long_consensus_case(77, 5000, 8000, 4000, 300_000, 0.0)
    eres, ores, hits = ec.run_both(rows, cs, rl, 7, 3, {}, [0], rd, batch_capacity=len(rd["pos"]), accum=eng.ACCUM_PARTITION)
    ec.assert_same(eres, ores, hits, False, n_rows)
    dres, _, _ = ec.run_both(rows, cs, rl, 7, 3, {}, [0], rd, batch_capacity=len(rd["pos"]), accum=eng.ACCUM_DEFAULT)
    for k in ("cnt", "cov", "cov_uniq"):
        assert np.array_equal(np.asarray(eres[k]), np.asarray(dres[k])), k
