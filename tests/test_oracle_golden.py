"""Pins oracle/ (our CPU restatement) against outputs of the reference itself.

golden inputs -> tests/refio.py parsers -> oracle/liboracle.so -> compare with the files the compiled
reference (oracle/_ref/iteres) wrote for the same inputs and options (tests/golden/*/manifest.json).
CPU only.
"""
import os

import numpy as np
import pytest

import goldencase as gc
import refio
from oracle import binding as orc


def _run_oracle(case, run):
    p = gc.parse_opts(run["cmd"], run["opts"])
    tm = gc.build_table_model(case, p["filter_field"], p["filter_name"])
    ot = orc.OracleTable(tm.chrom_size, tm.rep_len, len(tm.fams), len(tm.clas))
    st = ot.add_rows(tm.chrom, [r["start"] for r in tm.rows], [r["end"] for r in tm.rows],
                     [r["cons_start"] for r in tm.rows], [r["cons_end"] for r in tm.rows], tm.rep, tm.fam, tm.cla)
    assert (st == np.arange(len(tm.rows))).all()
    header, rd = gc.load_reads(case, run["aln"])

    def run_oracle(skip):
        return ot.run(p, gc.tid_map(header, tm, p["add_chr"]), rd["tid"], rd["pos"], rd["tmpend"], rd["mapq"], rd["flag"],
                      rd["mpos"], rd["isize"], skip=skip)
    # -R and the XA veto are string / file-order logic outside the C restatement: their Python restatement
    # (goldencase.side_masks) tells the oracle which records leave early, and owns the two counters they move
    dup, veto = gc.side_masks(p, tm, header, rd, ot, run_oracle)
    res = run_oracle(dup | veto)
    res["cnt"][11] -= np.uint64(int((dup & (rd["mapq"] >= p["mapq_min"])).sum()))
    res["cnt"][12] = np.uint64(int(veto.sum()))
    return p, tm, ot, rd, res


@pytest.mark.parametrize("case,run_name", gc.list_runs("stat"))
def test_stat_counts_and_coverage(case, run_name):
    run = gc.manifest_run(case, run_name)
    assert run["rc"] == 0
    p, tm, ot, rd, res = _run_oracle(case, run)
    gdir = os.path.join(gc.GOLDEN, case, run_name)
    rep = refio.parse_report(os.path.join(gdir, "out.iteres.report"))
    for k, v in rep.items():
        assert int(res["cnt"][k]) == v, f"cnt[{k}]"
    # subfamily.stat: counts per name + row order = kent hash order of names in insertion order
    _, rows = refio.parse_stat(os.path.join(gdir, "out.iteres.subfamily.stat"))
    assert [r[0] for r in rows] == refio.kent_hash_order(tm.names)
    glen = np.array([refio.u32(r["end"] - r["start"]) for r in tm.rows], np.uint64)
    for r in rows:
        i = tm.names.index(r[0])
        first = tm.rows[tm.rep_first_row[i]]
        assert (r[1], r[2]) == (first["fname"], first["cname"])
        assert int(r[3]) == int(tm.rep_len[i])
        assert int(r[4]) == int(res["rep_cnt"][i]), r[0]
        assert int(r[5]) == int(res["rep_cnt"][len(tm.names) + i]), r[0]
        assert int(r[6]) == int(glen[tm.rep == i].sum())
        assert int(r[7]) == int((tm.rep == i).sum())
    _, rows = refio.parse_stat(os.path.join(gdir, "out.iteres.family.stat"))
    assert [r[0] for r in rows] == refio.kent_hash_order(tm.fams)
    for r in rows:
        i = tm.fams.index(r[0])
        assert r[1] == tm.rows[tm.fam_first_row[i]]["cname"]
        assert (int(r[2]), int(r[3])) == (int(res["fam_cnt"][i]), int(res["fam_cnt"][len(tm.fams) + i]))
        assert int(r[4]) == int(glen[tm.fam == i].sum()) and int(r[5]) == int((tm.fam == i).sum())
    _, rows = refio.parse_stat(os.path.join(gdir, "out.iteres.class.stat"))
    assert [r[0] for r in rows] == refio.kent_hash_order(tm.clas)
    for r in rows:
        i = tm.clas.index(r[0])
        assert (int(r[1]), int(r[2])) == (int(res["cla_cnt"][i]), int(res["cla_cnt"][len(tm.clas) + i]))
    # per-base coverage
    for fn, key in (("out.iteres.wig", "cov"), ("out.iteres.unique.wig", "cov_uniq")):
        path = os.path.join(gdir, fn)
        if not (os.path.exists(path) or os.path.exists(path + ".gz")):
            continue
        wig, order = refio.parse_wig(path)
        assert order == [n for n in refio.kent_hash_order(tm.names) if tm.rep_len[tm.names.index(n)] != 0]
        for name, vec in wig.items():
            i = tm.names.index(name)
            got = res[key][int(ot.cov_off[i]): int(ot.cov_off[i + 1])]
            assert np.array_equal(got, vec), f"{fn} {name}"


@pytest.mark.parametrize("case,run_name", gc.list_runs("filter"))
def test_filter_loci(case, run_name):
    run = gc.manifest_run(case, run_name)
    assert run["rc"] == 0
    p, tm, ot, rd, res = _run_oracle(case, run)
    gdir = os.path.join(gc.GOLDEN, case, run_name)
    sub = p["filter_name"]
    rep = refio.parse_report(os.path.join(gdir, f"out_{sub}.iteres.reportloci"))
    for k, v in rep.items():
        assert int(res["cnt"][k]) == v, f"cnt[{k}]"
    _, rows = refio.parse_loci(os.path.join(gdir, f"out_{sub}.iteres.loci"))
    expect = []
    for k in gc.loci_row_order(tm):
        c = int(res["locus_cnt"][k])
        if c >= p["threshold"]:
            r = tm.rows[k]
            e = [r["chr"], str(r["start"]), str(r["end"]), str(refio.u32(r["end"] - r["start"])), r["name"], r["cname"], r["fname"], str(c)]
            if p["readlist"]:
                e.append(",".join(rd["qname"][i] for i in np.flatnonzero(res["hit_row"] == k)))
            expect.append(e)
    got = [r[:8] + ([r[10]] if p["readlist"] else []) for r in rows]
    assert got == expect


def test_find_order_quirk():
    """binKeeperFind's list order: coarse level first, bins descending, insertion ascending (binRange.c:209-225)."""
    ot = orc.OracleTable([600000], [100], 1, 1)
    # 131072 == 1 << 17 is the first finest-bin boundary
    rows = [(131000, 131200),   # spans the boundary      -> level 1 (bin 585)
            (131060, 131100),   # spans the boundary      -> level 1
            (131080, 131400),   # inside finest bin 1     -> bin 4682
            (131075, 131079),   # inside finest bin 1     -> bin 4682
            (130000, 131073),   # spans the boundary      -> level 1
            (131000, 131060)]   # inside finest bin 0     -> bin 4681
    n = len(rows)
    ot.add_rows([0] * n, [r[0] for r in rows], [r[1] for r in rows], [0] * n, [50] * n, [0] * n, [0] * n, [0] * n)
    assert list(ot.find(0, 131050, 131150)) == [0, 1, 4, 2, 3, 5]
    assert list(ot.find(0, 131077, 131078)) == [0, 1, 3]
    assert list(ot.find(0, -5, 130001)) == [4]          # start clipped to 0 (binRange.c:204)
    assert list(ot.find(0, 131300, 700000)) == [2]      # end clipped to chrom size (binRange.c:205)
    assert list(ot.find(0, 500, 500)) == []


def test_hash_model_matches_c():
    for s in ["AluY", "(TG)n", "L1PA2", "MER5A1", "x" * 40, "é"]:
        assert refio.kent_hash_string(s) == orc.hash_string(s)
