"""Host logic on CPU: the product's rmsk / size-file parser (iteres_amd/host/tables.c — pieces of the file parsed in
parallel, joined in file order) against the Python model of rmsk2binKeeperHash (tests/goldencase.py), for several
thread counts so that the pieces cut the file differently: same ids in the same first-appearance order, same
first-row family/class, same sums, same rows."""
import os
import subprocess

import numpy as np
import pytest

import goldencase as gc
import refio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "iteres_amd", "host")


@pytest.fixture(scope="module")
def dump(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bin") / "rmsk_dump")
    subprocess.check_call(["gcc", "-O2", "-g", "-fopenmp", "-std=gnu11", "-o", exe, os.path.join(HOST, "test", "rmsk_dump.c"),
                           os.path.join(HOST, "tables.c"), "-lm"])
    return exe


def model_text(tm: gc.TableModel):
    glen = np.array([refio.u32(r["end"] - r["start"]) for r in tm.rows], np.uint64)
    out = [f"seen\t{tm.n_seen}"]
    seen_chr = []
    for c in tm.chrom:
        if tm.chrom_names[c] not in seen_chr:
            seen_chr.append(tm.chrom_names[c])
    out += [f"chrom\t{n}\t{int(tm.chrom_size[tm.chrom_names.index(n)])}" for n in seen_chr]
    for i, n in enumerate(tm.names):
        first = tm.rows[tm.rep_first_row[i]]
        out.append(f"rep\t{n}\t{int(tm.rep_len[i])}\t{first['fname']}\t{first['cname']}\t{int((tm.rep == i).sum())}\t{int(glen[tm.rep == i].sum())}")
    for i, n in enumerate(tm.fams):
        out.append(f"fam\t{n}\t{tm.rows[tm.fam_first_row[i]]['cname']}\t{int((tm.fam == i).sum())}\t{int(glen[tm.fam == i].sum())}")
    for i, n in enumerate(tm.clas):
        out.append(f"cla\t{n}\t{int((tm.cla == i).sum())}\t{int(glen[tm.cla == i].sum())}")
    for k, r in enumerate(tm.rows):
        out.append(f"row\t{r['chr']}\t{int(tm.chrom[k])}\t{refio.u32(r['start'])}\t{refio.u32(r['end'])}\t{refio.u32(r['cons_start'])}\t"
                   f"{refio.u32(r['cons_end'])}\t{r['name']}\t{r['fname']}\t{r['cname']}")
    return out


@pytest.mark.parametrize("case,filt", [("quirks", None), ("mid", None), ("mid", (12, None)), ("manynames", None), ("sidechan", None)])
def test_rmsk_parse_matches_model(case, filt, dump, tmp_path):
    src = os.path.join(gc.GOLDEN, case, "in")
    paths = [refio.materialise(src, n, str(tmp_path)) for n in ("chrom.sizes", "rep.sizes", "rmsk.txt")]
    extra = []
    ff, fn = 0, "ALL"
    if filt:
        tm0 = gc.build_table_model(case)
        ff, fn = filt[0], tm0.fams[3]                     # a family filter (filter -f)
        extra = [str(ff), fn]
    want = model_text(gc.build_table_model(case, ff, fn))
    for threads in (1, 3, 8):
        pr = subprocess.run([dump] + paths + extra, capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS=str(threads)))
        assert pr.returncode == 0, pr.stderr
        got = pr.stdout.rstrip("\n").split("\n")
        assert got == want, (threads, next((a, b) for a, b in zip(got, want) if a != b))


def test_compressed_inputs_with_awkward_names(dump, tmp_path):
    """.gz inputs go through a decompressor that is exec'd with the path as its own argument (no shell): a name with a
    quote, a space or a leading dash reads like any other (the reference's pipeline does the same, cuskent/pipeline.c)."""
    import gzip
    import shutil
    src = os.path.join(gc.GOLDEN, "quirks", "in")
    plain = [refio.materialise(src, n, str(tmp_path)) for n in ("chrom.sizes", "rep.sizes", "rmsk.txt")]
    want = subprocess.run([dump] + plain, capture_output=True, text=True)
    assert want.returncode == 0
    odd = tmp_path / "it's a dir"
    odd.mkdir()
    zpaths = []
    for p, name in zip(plain, ("chrom's.sizes.gz", "rep sizes.gz", "-rmsk.txt.gz")):
        z = odd / name
        with open(p, "rb") as fi, gzip.open(z, "wb") as fo:
            shutil.copyfileobj(fi, fo)
        zpaths.append(name)
    got = subprocess.run([dump] + zpaths, capture_output=True, text=True, cwd=str(odd))
    assert got.returncode == 0, got.stderr
    assert got.stdout == want.stdout
