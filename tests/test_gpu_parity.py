"""GPU parity tests proper: the HIP engine, called through the C ABI, against the oracle — bit-exact.

Run on the MI355X box with `pytest -m gpu`. They compare (a) every golden case/option set, (b) seeded
random workloads (sorted, unsorted, paired, pathological overlap), (c) edge cases, and (d) at larger
sizes, size-independent properties (batch-split invariance, permutation invariance, both accumulate paths).
"""
import numpy as np
import pytest

import enginecase as ec
import goldencase as gc
from iteres_amd import engine as eng
from iteres_amd import synth

pytestmark = pytest.mark.gpu

import os
ACCUMS = [int(x) for x in os.environ.get("ITX_TEST_ACCUMS", "1,2").split(",")]


@pytest.mark.parametrize("case,run_name", [r for r in gc.list_runs("stat") + gc.list_runs("filter") if r[0] != "manynames"])
@pytest.mark.parametrize("accum", ACCUMS)
def test_golden_case(case, run_name, accum):
    run = gc.manifest_run(case, run_name)
    p = gc.parse_opts(run["cmd"], run["opts"])
    tm = gc.build_table_model(case, p["filter_field"], p["filter_name"])
    header, rd = gc.load_reads(case, run["aln"])
    rows = ec.table_from_model(tm)
    eres, ores, hits = ec.run_both(rows, tm.chrom_size, tm.rep_len, len(tm.fams), len(tm.clas), p,
                                   gc.tid_map(header, tm, p["add_chr"]), rd, batch_capacity=7001, accum=accum)
    ec.assert_same(eres, ores, hits, p["filter_mode"], len(rows))


def _synth_case(seed, n_iv, n_reads, chroms, paired=0.0, sorted_=True, overlap=0.05, **kw):
    t = synth.make_table(seed, chroms, n_iv, n_names=kw.get("n_names", 500), n_fams=40, n_clas=12, overlap_frac=overlap,
                         shuffle_frac=0.03, inconsistent_frac=0.01)
    names_len = np.array([t.rep_len.get(n, 0) for n in t.names], np.uint32)
    rows = eng.make_rows(t.chrom, t.start, t.end, t.cons_start, t.cons_end, t.rep_name, t.fam_of_row, t.cla_of_row)
    header = list(chroms) + [("chrNotInSizes", 5000)]
    r = synth.make_reads(seed + 1, header, n_reads, read_len=(30, 150), paired_frac=paired, sorted_=sorted_, odd_cigar_frac=0.1)
    rd = {"tid": r.tid, "pos": r.pos, "tmpend": r.tmpend(), "mapq": r.mapq, "flag": r.flag, "mpos": r.mpos, "isize": r.isize}
    tid2chrom = np.array(list(range(len(chroms))) + [-1], np.int32)
    return rows, np.array([s for _, s in chroms], np.int64), names_len, len(t.fams), len(t.clas), tid2chrom, rd


@pytest.mark.parametrize("accum", ACCUMS)
@pytest.mark.parametrize("variant", ["sorted_se", "unsorted_se", "paired", "paired_T", "paired_D_E0", "filter"])
def test_random_workloads(variant, accum):
    chroms = [("c1", 40_000_000), ("c2", 9_000_000), ("c3", 300_000)]
    paired = 0.5 if variant.startswith("paired") else 0.0
    rows, cs, rl, nf, nc, t2c, rd = _synth_case(7, 60_000, 120_000, chroms, paired=paired, sorted_=(variant != "unsorted_se"))
    p = dict(mapq_min=10, min_cov=0.0001, extension=150, isize_max=500)
    if variant == "paired_T": p["treat_pe_as_se"] = True
    if variant == "paired_D_E0": p.update(discard_half_mapped=True, extension=0, min_cov=0.25, mapq_min=30)
    if variant == "filter": p["filter_mode"] = True
    eres, ores, hits = ec.run_both(rows, cs, rl, nf, nc, p, t2c, rd, batch_capacity=33_333, accum=accum)
    ec.assert_same(eres, ores, hits, bool(p.get("filter_mode")), len(rows))
    assert int(ores["cnt"][9]) > 10_000


@pytest.mark.parametrize("accum", ACCUMS)
def test_pathological_overlap(accum):
    """Deeply nested / heavily overlapping rows across bin levels: many hits per read, so the replay of the
    reference's best-hit rule through list-order ranks is exercised with n >> 2."""
    rng = np.random.default_rng(5)
    size = 3_000_000
    n = 4000
    centre = rng.integers(100_000, size - 100_000, n)
    half = (2 ** rng.uniform(3, 17, n)).astype(np.int64)           # 8 bp .. 128 kb
    start = np.maximum(centre - half, 0)
    end = np.minimum(centre + half, size)
    rep = rng.integers(0, 50, n)
    rows = eng.make_rows(np.zeros(n, np.int32), start, end, rng.integers(0, 300, n), rng.integers(100, 900, n), rep, rep % 7, rep % 3)
    rl = rng.integers(0, 800, 50).astype(np.uint32)
    m = 50_000
    pos = np.sort(rng.integers(0, size, m)).astype(np.int32)
    rd = {"tid": np.zeros(m, np.int32), "pos": pos, "tmpend": (pos + rng.integers(20, 400, m)).astype(np.int32),
          "mapq": rng.integers(0, 61, m).astype(np.uint8), "flag": np.where(rng.random(m) < 0.5, 16, 0).astype(np.uint16),
          "mpos": np.zeros(m, np.int32), "isize": np.zeros(m, np.int32)}
    for E in (0, 150):
        p = dict(extension=E, min_cov=0.0001)
        eres, ores, hits = ec.run_both(rows, [size], rl, 7, 3, p, [0], rd, accum=accum)
        ec.assert_same(eres, ores, hits, False, n)
    assert (ores["hit_row"] >= 0).mean() > 0.5


@pytest.mark.parametrize("accum", ACCUMS)
def test_edge_cases(accum):
    size = 1_000_000
    rows = eng.make_rows([0, 0, 0], [0, 500, 999_000], [400, 500, 1_000_000], [0, 0, 4294967291], [400, 10, 600],
                         [0, 1, 0], [0, 0, 0], [0, 0, 0])               # includes an empty row (500,500) and a huge cons_start
    rl = np.array([600, 0], np.uint32)
    cs = [size, 2]                                                     # chromosome 1 has the "missing" size 2
    tid = np.array([0, 0, 0, 0, 5, -1, 0, 0, 1, 0], np.int32)          # tid 5 / -1: outside the header
    pos = np.array([0, 399, 999_900, 2_000_000, 10, 10, -1, 499, 10, 999_995], np.int32)
    tmpend = pos + np.array([50, 50, 500, 50, 50, 50, 50, 3, 5, 3], np.int32)
    rd = {"tid": tid, "pos": pos, "tmpend": tmpend, "mapq": np.full(10, 30, np.uint8),
          "flag": np.array([0, 16, 0, 0, 0, 0, 0, 0, 0, 16], np.uint16), "mpos": np.zeros(10, np.int32), "isize": np.zeros(10, np.int32)}
    for E in (0, 150, 4_000_000_000):
        p = dict(extension=E)
        eres, ores, hits = ec.run_both(rows, cs, rl, 1, 1, p, [0, 1], rd, accum=accum)
        ec.assert_same(eres, ores, hits, False, 3)
    # empty batch and single record
    t = eng.Table(rows, cs, rl, 1, 1)
    e = eng.Engine(t, dict(accum=accum), batch_capacity=16)
    e.set_tidmap([0, 1])
    z = np.zeros(0, np.int32)
    e.submit_host(z, z, z, z.astype(np.uint8), z.astype(np.uint8))
    r0 = e.finish()
    assert r0["cnt"].sum() == 0 and r0["cov"].sum() == 0
    e.close(); t.close()


def test_f32_ratio_exactness():
    """generic.c:296-301 compares float ratios: with a huge fragment (-E 0 and an N-skipping CIGAR) the f32
    rounding of overlap/length decides the threshold test; the device must round like the CPU."""
    rng = np.random.default_rng(11)
    size = 400_000_000
    n = 3000
    start = np.sort(rng.integers(0, size - 50_000, n))
    end = start + rng.integers(1, 40_000, n)
    rep = rng.integers(0, 10, n)
    rows = eng.make_rows(np.zeros(n, np.int32), start, end, np.zeros(n), rng.integers(1, 900, n), rep, rep % 3, rep % 2)
    rl = np.full(10, 500, np.uint32)
    m = 40_000
    pos = rng.integers(0, size - 40_000_000, m).astype(np.int32)
    ln = rng.integers(1, 2**25, m)
    rd = {"tid": np.zeros(m, np.int32), "pos": pos, "tmpend": (pos + ln).astype(np.int32), "mapq": np.full(m, 40, np.uint8),
          "flag": np.zeros(m, np.uint16), "mpos": np.zeros(m, np.int32), "isize": np.zeros(m, np.int32)}
    for mc in (1e-4, 0.001, 0.0123456):
        eres, ores, hits = ec.run_both(rows, [size], rl, 3, 2, dict(extension=0, min_cov=mc), [0], rd)
        ec.assert_same(eres, ores, hits, False, n)
    assert 0.02 < (ores["hit_row"] >= 0).mean() < 0.5       # the last threshold really cuts


def test_min_cov_band():
    """Quotients within 2^-20 of -c leave the one-multiply test (itx_cov_bounds) for the exact division: overlaps of
    qlen/2 + d against -c 0.5 with qlen = 2^22 sit 2^-22 apart around the threshold."""
    size = 100_000_000
    qlen = 1 << 22
    R = 50_000_000
    rows = eng.make_rows([0], [R], [R + 20_000_000], [0], [500], [0], [0], [0])
    rl = np.array([500], np.uint32)
    d = np.arange(-40, 41)
    pos = (R - (qlen // 2 - d)).astype(np.int32)
    m = len(pos)
    rd = {"tid": np.zeros(m, np.int32), "pos": pos, "tmpend": (pos + qlen).astype(np.int32), "mapq": np.full(m, 40, np.uint8),
          "flag": np.zeros(m, np.uint16), "mpos": np.zeros(m, np.int32), "isize": np.zeros(m, np.int32)}
    for mc in (0.5, 0.5000001, 0.4999999, 0.50000006):
        eres, ores, hits = ec.run_both(rows, [size], rl, 1, 1, dict(extension=0, min_cov=mc), [0], rd)
        ec.assert_same(eres, ores, hits, False, 1)
        assert 0 < (ores["hit_row"] >= 0).sum() < m          # the threshold falls inside the sweep


@pytest.mark.parametrize("accum", ACCUMS)
def test_nolookup_and_veto(accum):
    """ITX_F5_NOLOOKUP: records a caller takes out between the mapped-read counters and the lookup (-R duplicates,
    marked up front; XA vetoes, marked after a classify-only pass over the slot) — counted, never accumulated."""
    chroms = [("c1", 40_000_000), ("c2", 9_000_000)]
    rows, cs, rl, nf, nc, t2c, rd = _synth_case(77, 60_000, 300_000, chroms, paired=0.3)
    rng = np.random.default_rng(79)
    skip = rng.random(len(rd["tid"])) < 0.2
    veto = lambda h: (h >= 0) & (h % 3 == 0)
    for kw in (dict(skip=skip), dict(veto=veto), dict(skip=skip, veto=veto)):
        eres, ores, hits = ec.run_both(rows, cs, rl, nf, nc, dict(), t2c, rd, batch_capacity=50_001, accum=accum, **kw)
        ec.assert_same(eres, ores, hits, False, len(rows))
    assert (ores["hit_row"] >= 0).sum() > 10_000


def test_first_hit_lookup():
    """itx_engine_first_hit_*: the first row binKeeperFind returns for a plain interval (the cpg commands' lookup,
    generic.c:1082-1088) — sorted CpG-like sites, unsorted intervals of every size, chromosome edges, nested rows."""
    chroms = [("c1", 30_000_000), ("c2", 6_000_000), ("c3", 400_000)]
    rows, cs, rl, nf, nc, t2c, _ = _synth_case(91, 80_000, 10, chroms, overlap=0.15)
    ot = ec.oracle_table(rows, cs, rl, nf, nc)
    rng = np.random.default_rng(92)
    t = eng.Table(rows, cs, rl, nf, nc)
    e = eng.Engine(t, dict(), batch_capacity=70_001)
    e.set_tidmap(list(t2c))
    m = 200_000
    # CpG-like: sorted 2-bp sites; then unsorted intervals from 1 bp to 300 kb, some hanging over the chromosome ends
    tid = np.sort(rng.integers(0, 3, m)).astype(np.int32)
    pos = np.zeros(m, np.int64)
    for c in range(3):
        k = tid == c
        pos[k] = np.sort(rng.integers(0, int(cs[c]) - 1, int(k.sum())))
    cases = [(tid, pos.astype(np.int32), (pos + 2).astype(np.int32))]
    tid2 = rng.integers(-1, 5, m).astype(np.int32)                       # -1 and 4: no such chromosome; 3 = chrNotInSizes
    size2 = np.array([int(cs[c]) if 0 <= c < 3 else 5000 for c in tid2])
    st2 = (rng.random(m) * (size2 + 2000) - 1000).astype(np.int64)
    ln2 = (2 ** rng.uniform(0, 18, m)).astype(np.int64)
    cases.append((tid2, st2.astype(np.int32), (st2 + ln2).astype(np.int32)))
    for tid_, s_, e_ in cases:
        got = e.first_hits_host(tid_, s_, e_)
        idx = rng.choice(m, 6000, replace=False)                         # the oracle's find is a Python-level call: sample
        for i in idx:
            c = int(t2c[tid_[i]]) if 0 <= tid_[i] < len(t2c) else -1
            h = ot.find(c, int(s_[i]), int(e_[i])) if c >= 0 else []
            assert int(got[i]) == (int(h[0]) if len(h) else -1), (i, int(tid_[i]), int(s_[i]), int(e_[i]))
        assert (got >= 0).mean() > 0.2
    e.close(); t.close(); ot.close()


@pytest.mark.parametrize("accum", ACCUMS)
def test_unit_and_window_shapes(accum):
    """Slot-space shapes the partition path has to get right: thousands of tiny units (consensus length 0, 1, 2 — many
    units inside one 8192-slot window, empty ones included), units longer than a window, rows whose consensus range
    crosses window boundaries (start and end marks in different partitions: two keys), reads far longer than a row."""
    rng = np.random.default_rng(123)
    size = 40_000_000
    n, n_names = 120_000, 3000
    start = np.sort(rng.integers(0, size - 30_000, n))
    glen = (2 ** rng.uniform(3, 14, n)).astype(np.int64)               # 8 bp .. 16 kb
    end = np.minimum(start + glen, size)
    rep = rng.integers(0, n_names, n)
    lens = rng.choice(np.array([0, 0, 1, 1, 2, 3, 5, 40, 300, 8191, 8192, 8193, 20_000], np.int64), n_names)
    rl = lens.astype(np.uint32)
    cons_start = (rng.random(n) * np.maximum(lens[rep] - 1, 1)).astype(np.int64)
    cons_end = cons_start + rng.integers(0, 30_000, n)
    rows = eng.make_rows(np.zeros(n, np.int32), start, end, cons_start, cons_end, rep, rep % 50, rep % 9)
    m = 400_000
    pos = np.sort(rng.integers(0, size - 1, m)).astype(np.int32)
    ln = np.where(rng.random(m) < 0.7, rng.integers(30, 200, m), (2 ** rng.uniform(5, 15, m)).astype(np.int64))
    rd = {"tid": np.zeros(m, np.int32), "pos": pos, "tmpend": (pos + ln).astype(np.int32), "mapq": rng.integers(0, 61, m).astype(np.uint8),
          "flag": np.where(rng.random(m) < 0.5, 16, 0).astype(np.uint16), "mpos": np.zeros(m, np.int32), "isize": np.zeros(m, np.int32)}
    for E in (0, 150):
        eres, ores, hits = ec.run_both(rows, [size], rl, 50, 9, dict(extension=E), [0], rd, batch_capacity=150_001, accum=accum)
        ec.assert_same(eres, ores, hits, False, n)
    assert (ores["hit_row"] >= 0).mean() > 0.3 and int(ores["cov"].astype(np.uint64).sum()) > 1_000_000
    if accum == eng.ACCUM_PARTITION:
        # a slot space beyond 4096 windows of 8192 (210 M slots): the partitions become 2^16 slots wide, k_hist walks their
        # eight windows in turn; same sums as the oracle, asked for explicitly and by default
        big = np.full(n_names, 70_000, np.uint32)
        k = 60_000
        sub = {key: v[:k] for key, v in rd.items()}
        for how in (eng.ACCUM_PARTITION, eng.ACCUM_DEFAULT):
            eres, ores, hits = ec.run_both(rows, [size], big, 50, 9, dict(), [0], sub, batch_capacity=20_000, accum=how)
            ec.assert_same(eres, ores, hits, False, n)
        # beyond 4096 x 2^16 slots the partition path is refused when asked for explicitly (the default then takes the
        # atomics path: same sums, slower)
        huge = np.full(n_names, 90_000, np.uint32)                      # 270 M slots
        t = eng.Table(rows, [size], huge, 50, 9)
        with pytest.raises(eng.ItxError):
            eng.Engine(t, dict(accum=eng.ACCUM_PARTITION), batch_capacity=1000)
        e = eng.Engine(t, dict(accum=eng.ACCUM_DEFAULT), batch_capacity=1000)
        e.close()
        t.close()


def test_long_spans_few_workgroups(monkeypatch):
    """Few workgroups with long spans (ITX_STREAM_BLOCKS): the last workgroup's span is mostly empty, its waves still
    write their keys into their own quarters of the region — the key buffer has to reach that far."""
    chroms = [("c1", 20_000_000)]
    rows, cs, rl, nf, nc, t2c, rd = _synth_case(55, 40_000, 100_000, chroms)
    for blocks, cap in (("8", 100_000), ("3", 100_000), ("8", 33_333)):
        monkeypatch.setenv("ITX_STREAM_BLOCKS", blocks)
        eres, ores, hits = ec.run_both(rows, cs, rl, nf, nc, dict(), t2c, rd, batch_capacity=cap, accum=eng.ACCUM_PARTITION)
        ec.assert_same(eres, ores, hits, False, len(rows))


def test_properties_at_scale():
    """2 M reads vs 300 k rows: results must not depend on how the stream is cut into batches, on record order,
    or on the accumulate path (all sums are integer and commutative)."""
    chroms = [("c1", 150_000_000), ("c2", 60_000_000)]
    rows, cs, rl, nf, nc, t2c, _ = _synth_case(31, 300_000, 10, chroms)
    tid, pos, tmpend, mapq, f5 = synth.make_reads_soa(32, chroms, 2_000_000)
    t = eng.Table(rows, cs, rl, nf, nc)
    outs = []
    perm = np.random.default_rng(1).permutation(len(tid))
    A0, A1 = ACCUMS[0], ACCUMS[-1]
    for accum, cap, order in ((A0, 1 << 21, None), (A1, 1 << 21, None), (A1, 77_777, None), (A1, 1 << 19, perm)):
        e = eng.Engine(t, dict(accum=accum), batch_capacity=cap)
        e.set_tidmap(t2c[:2])
        a = [x if order is None else x[order] for x in (tid, pos, tmpend, mapq, f5)]
        e.submit_host(*a)
        outs.append(e.finish())
        e.close()
    for o in outs[1:]:
        for k in ec.KEYS:
            assert np.array_equal(o[k], outs[0][k]), k
    r = outs[0]
    assert int(r["rep_cnt"][: len(rl)].sum()) == int(r["cnt"][9]) == int(r["fam_cnt"][:nf].sum()) == int(r["cla_cnt"][:nc].sum())
    assert int(r["rep_cnt"][len(rl):].sum()) == int(r["cnt"][10])
    t.close()
