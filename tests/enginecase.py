"""Helpers shared by the GPU parity tests: run the same inputs through the HIP engine (via the C ABI) and
through the oracle, and compare every output array."""
from __future__ import annotations

import numpy as np

import goldencase as gc
from iteres_amd import engine as eng
from oracle import binding as orc

KEYS = ("cnt", "rep_cnt", "fam_cnt", "cla_cnt", "cov", "cov_uniq", "locus_cnt")


def table_from_model(tm: gc.TableModel):
    rows = eng.make_rows(tm.chrom, [r["start"] for r in tm.rows], [r["end"] for r in tm.rows],
                         [r["cons_start"] for r in tm.rows], [r["cons_end"] for r in tm.rows], tm.rep, tm.fam, tm.cla)
    return rows


def oracle_table(rows, chrom_size, rep_len, n_fam, n_cla):
    ot = orc.OracleTable(chrom_size, rep_len, n_fam, n_cla)
    st = ot.add_rows(rows["chrom"], rows["start"], rows["end"], rows["cons_start"], rows["cons_end"], rows["rep"], rows["fam"], rows["cla"])
    assert (st == np.arange(len(rows))).all()
    return ot


def run_both(rows, chrom_size, rep_len, n_fam, n_cla, params, tid2chrom, rd, batch_capacity=1 << 16, accum=eng.ACCUM_DEFAULT,
             via="host", skip=None, veto=None):
    """rd: dict tid,pos,tmpend,mapq,flag(16-bit BAM flags),mpos,isize. Returns (engine_result, oracle_result, hits_e).
    skip: bool mask of records the caller marks ITX_F5_NOLOOKUP up front (a -R duplicate); veto(hit_rows) -> bool mask:
    records marked after a classify-only pass (an XA veto). The oracle gets the union as its `skip`."""
    ot = oracle_table(rows, chrom_size, rep_len, n_fam, n_cla)
    args = (params, tid2chrom, rd["tid"], rd["pos"], rd["tmpend"], rd["mapq"], rd["flag"], rd.get("mpos"), rd.get("isize"))
    oskip = None if skip is None else np.asarray(skip, bool).copy()
    if veto is not None:
        first = ot.run(*args, skip=oskip)
        v = np.asarray(veto(first["hit_row"]), bool)
        oskip = v if oskip is None else (oskip | v)
    ores = ot.run(*args, skip=oskip)
    ot.close()
    t = eng.Table(rows, chrom_size, rep_len, n_fam, n_cla)
    p = dict(params)
    p["accum"] = accum
    e = eng.Engine(t, p, batch_capacity=batch_capacity)
    e.set_tidmap(tid2chrom)
    f5 = eng.flag5(rd["flag"])
    if skip is not None:
        f5 = f5 | (np.asarray(skip, bool).astype(np.uint8) * np.uint8(eng.F5_NOLOOKUP))
    paired = bool((np.asarray(rd["flag"]) & 1).any())
    hits = e.submit_host(rd["tid"], rd["pos"], rd["tmpend"], rd["mapq"], f5, rd["mpos"] if paired else None,
                         rd["isize"] if paired else None, want_hits=True,
                         veto=None if veto is None else (lambda off, h: veto(h.astype(np.int64))))
    eres = e.finish()
    e.close()
    t.close()
    return eres, ores, hits


def assert_same(eres, ores, hits, filter_mode, n_rows):
    assert np.array_equal(hits.astype(np.int64), ores["hit_row"]), "hit rows differ"
    # cnt[8] (-R) and cnt[12] (XA veto) are host-side in the product and zero in both
    assert np.array_equal(eres["cnt"], ores["cnt"]), (eres["cnt"], ores["cnt"])
    if filter_mode:
        assert np.array_equal(eres["locus_cnt"][:n_rows], ores["locus_cnt"][:n_rows])
    else:
        for k in ("rep_cnt", "fam_cnt", "cla_cnt", "cov", "cov_uniq"):
            assert np.array_equal(eres[k], ores[k]), k
