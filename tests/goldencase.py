"""Shared test-side model of a golden case: inputs parsed in Python (tests/refio.py), the table
bookkeeping of generic.c:1578-1707 restated on the test side, option parsing of stat.c:46-68 /
filter.c:46-68, and helpers that turn accumulator arrays into the quantities the reference's output
files hold."""
from __future__ import annotations

import json
import os
from dataclasses import dataclass

import numpy as np

import refio

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def list_runs(cmd=None):
    out = []
    for case in sorted(os.listdir(GOLDEN)):
        mf = os.path.join(GOLDEN, case, "manifest.json")
        if not os.path.exists(mf):
            continue
        for run in json.load(open(mf))["runs"]:
            if cmd is None or run["cmd"] == cmd:
                out.append((case, run["name"]))
    return out


def manifest_run(case, run_name):
    man = json.load(open(os.path.join(GOLDEN, case, "manifest.json")))
    return next(r for r in man["runs"] if r["name"] == run_name)


def parse_opts(cmd, opts):
    """-> dict of engine parameters + output options, defaults from stat.c:34-36 / filter.c:37-41."""
    p = dict(mapq_min=10, min_cov=0.0001, extension=150, isize_max=500, treat_pe_as_se=False,
             discard_half_mapped=False, filter_mode=(cmd == "filter"), sam=False, add_chr=False, norm=0, norm2=0,
             threshold=1, readlist=False, filter_field=0, filter_name="ALL", keep_wig=False, xa_off=False, dedup=False,
             bed=False, bed_uniq=False)
    i = 0
    while i < len(opts):
        o = opts[i]
        arg = opts[i + 1] if i + 1 < len(opts) else None
        if o == "-S": p["sam"] = True
        elif o == "-Q": p["mapq_min"] = int(arg); i += 1
        elif o == "-E": p["extension"] = int(arg); i += 1
        elif o == "-I": p["isize_max"] = int(arg); i += 1
        elif o == "-T": p["treat_pe_as_se"] = True
        elif o == "-D": p["discard_half_mapped"] = True
        elif o == "-C": p["add_chr"] = True
        elif o == "-w": p["keep_wig"] = True
        elif o == "-x": p["xa_off"] = True
        elif o == "-R": p["dedup"] = True
        elif o == "-B": p["bed"] = True
        elif o == "-V": p["bed_uniq"] = True
        elif o == "-N": p["norm"] = int(arg); i += 1
        elif o == "-U": p["norm2"] = int(arg); i += 1
        elif o == "-t": p["threshold"] = int(arg); i += 1
        elif o == "-r": p["readlist"] = True
        elif o == "-g" or (o == "-c" and cmd == "stat"): p["min_cov"] = float(arg); i += 1
        elif o == "-n": p["filter_field"], p["filter_name"] = 10, arg; i += 1
        elif o == "-c": p["filter_field"], p["filter_name"] = 11, arg; i += 1
        elif o == "-f": p["filter_field"], p["filter_name"] = 12, arg; i += 1
        else: raise ValueError(o)
        i += 1
    if p["filter_name"] == "ALL":
        p["filter_field"] = 0
    return p


@dataclass
class TableModel:
    """What rmsk2binKeeperHash leaves behind (generic.c:1578-1707), as flat arrays."""
    chrom_names: list
    chrom_size: np.ndarray
    rows: list                 # kept rows (dicts from refio.read_rmsk) in file order
    chrom: np.ndarray
    rep: np.ndarray
    fam: np.ndarray
    cla: np.ndarray
    names: list
    fams: list
    clas: list
    rep_len: np.ndarray
    rep_first_row: list        # first kept row index per name (its fname/cname are the ones reported)
    fam_first_row: list
    n_seen: int                # rows counted in the "Total %d repeats" banner


def build_table_model(case, filter_field=0, filter_name="ALL") -> TableModel:
    d = os.path.join(GOLDEN, case, "in")
    sizes = refio.read_sizes(os.path.join(d, "chrom.sizes"))
    repsz = refio.read_sizes(os.path.join(d, "rep.sizes"))
    chrom_names = list(sizes)
    cidx = {n: i for i, n in enumerate(chrom_names)}
    rows, chrom, rep, fam, cla = [], [], [], [], []
    names, fams, clas, ni, fi, ci = [], [], [], {}, {}, {}
    rep_first, fam_first = [], []
    n_seen = 0
    for r in refio.read_rmsk(os.path.join(d, "rmsk.txt")):
        if filter_field and r["cols"][filter_field] != filter_name:
            continue
        n_seen += 1
        # hashIntValDefault(chrHash, chr, 0) == 0 -> row dropped (generic.c:1618-1622)
        if sizes.get(r["chr"], 0) == 0:
            continue
        k = len(rows)
        rows.append(r)
        chrom.append(cidx[r["chr"]])
        for key, lst, dct, first in ((r["name"], names, ni, rep_first), (r["fname"], fams, fi, fam_first), (r["cname"], clas, ci, None)):
            if key not in dct:
                dct[key] = len(lst)
                lst.append(key)
                if first is not None:
                    first.append(k)
        rep.append(ni[r["name"]]); fam.append(fi[r["fname"]]); cla.append(ci[r["cname"]])
    rep_len = np.array([refio.u32(repsz.get(n, 0)) for n in names], np.uint32)
    return TableModel(chrom_names, np.array([sizes[n] for n in chrom_names], np.int64), rows,
                      np.array(chrom, np.int32), np.array(rep, np.int32), np.array(fam, np.int32), np.array(cla, np.int32),
                      names, fams, clas, rep_len, rep_first, fam_first, n_seen)


def rename_chr(name, add_chr):
    """generic.c:781-791. Returns None when -C drops the record ("GL*")."""
    if not add_chr:
        return name
    if name.startswith("GL"):
        return None
    if name.upper() == "MT":
        return "chrM"
    if not name.startswith("chr"):
        return "chr" + name
    return name


def tid_map(header, tm: TableModel, add_chr=False):
    cidx = {n: i for i, n in enumerate(tm.chrom_names)}
    out = []
    for name, _ in header:
        nm = rename_chr(name, add_chr)
        if nm is None:
            out.append(-2)
        elif nm in cidx and int(tm.chrom_size[cidx[nm]]) != 2:
            out.append(cidx[nm])
        else:
            out.append(-1)
    return np.array(out, np.int32)


def load_reads(case, aln):
    path = os.path.join(GOLDEN, case, "in", aln)
    return refio.read_sam(path) if aln.endswith(".sam") else refio.read_bam(path)


# cuskent/binRange.c:119-138
_OFFS = (4681, 585, 73, 9, 1, 0)


def bin_of(start, end):
    s, e = start >> 17, (end - 1) >> 17
    for off in _OFFS:
        if s == e:
            return off + s
        s >>= 3
        e >>= 3
    raise ValueError


def loci_row_order(tm: TableModel):
    """Row order of writeFilterOut (generic.c:1719-1723): chrom hash order x binKeeperNext order
    (bin ascending, newest insertion first; cuskent/binRange.c:365-392)."""
    first_seen = []
    for c in tm.chrom:
        n = tm.chrom_names[c]
        if n not in first_seen:
            first_seen.append(n)
    order = []
    for cname in refio.kent_hash_order(first_seen):
        ci = tm.chrom_names.index(cname)
        idx = [k for k in range(len(tm.rows)) if tm.chrom[k] == ci]
        idx.sort(key=lambda k: (bin_of(tm.rows[k]["start"], tm.rows[k]["end"]), -k))
        order += idx
    return order


# ---------------------------------------------------------------------------------------------------------------
# The string / file-order parts of the reference's record loop, restated in Python for the small golden cases
# (test infrastructure, like oracle/): which records leave through the -R `continue` (generic.c:907-919) and
# through the XA veto (generic.c:303-341, 972-982). Returned as masks for the oracle's `skip`.

def derive_py(p, tid2chrom, chrom_size, rd, i):
    """generic.c:764-905 for record i -> (start, end, strand) as the loop holds them, or None when it is skipped."""
    fl = int(rd["flag"][i])
    if fl & refio.FUNMAP:
        return None
    t = int(rd["tid"][i])
    c = int(tid2chrom[t]) if 0 <= t < len(tid2chrom) else -1
    if c < 0:
        return None
    cend = refio.u32(int(chrom_size[c]) - 1)
    if cend == 1:
        return None
    pos, tmpend, isz, mpos = int(rd["pos"][i]), int(rd["tmpend"][i]), int(rd["isize"][i]), int(rd["mpos"][i])
    E = p["extension"]
    se = True
    if not p["treat_pe_as_se"] and (fl & 0x1):
        if not (fl & 0x8):
            if not (fl & 0x40) or abs(isz) > p["isize_max"] or isz == 0:
                return None
            se = False
        elif p["discard_half_mapped"]:
            return None
    if se:
        start, end = refio.u32(pos), min(cend, refio.u32(tmpend))
        strand = "-" if fl & 0x10 else "+"
        if E:
            if strand == "+":
                end = min(refio.u32(start + E), cend)
            else:
                start = 0 if end < E else end - E
    elif isz > 0:
        start = refio.u32(pos)
        end, strand = min(cend, refio.u32(start + isz)), "+"
    else:
        start = refio.u32(mpos)
        end, strand = min(cend, refio.u32(start - isz)), "-"
    return start, end, strand


def side_masks(p, tm, header, rd, ot, run_oracle):
    """-> (dup_mask, veto_mask). run_oracle(skip) -> oracle result (for the chosen rows the veto looks at)."""
    n = len(rd["tid"])
    t2c = tid_map(header, tm, p["add_chr"])
    names = [rename_chr(nm, p["add_chr"]) for nm, _ in header]
    iv = [derive_py(p, t2c, tm.chrom_size, rd, i) for i in range(n)]
    dup = np.zeros(n, bool)
    if p["dedup"]:
        seen, key = set(), None          # the reference's key buffer starts uninitialised; the goldens begin with MAPQ >= Q
        for i in range(n):
            if iv[i] is None:
                continue
            if int(rd["mapq"][i]) >= p["mapq_min"]:
                key = "%s:%u:%u:%s" % (names[int(rd["tid"][i])], iv[i][0], iv[i][1], iv[i][2])
            if key in seen:
                dup[i] = True
            else:
                seen.add(key)
    veto = np.zeros(n, bool)
    if not p["filter_mode"] and not p["xa_off"] and rd.get("xa") is not None and any(x is not None for x in rd["xa"]):
        first = run_oracle(dup)
        cidx = {nm: i for i, nm in enumerate(tm.chrom_names)}
        has_rows = set(int(c) for c in tm.chrom)
        for i in range(n):
            xa = rd["xa"][i]
            row = int(first["hit_row"][i])
            if xa is None or iv[i] is None or dup[i] or row < 0:
                continue
            chosen = tm.rows[row]["name"].upper()
            qlen = refio.u32(iv[i][1] - iv[i][0])
            qlen = qlen - (1 << 32) if qlen >= (1 << 31) else qlen                       # (int)qlen
            fields = xa.split(";")[:100] if xa else []                                  # chopByChar, 100 slots
            for alt in fields:
                if not alt:
                    continue
                w = alt.split(",")
                assert len(w) >= 4
                if refio.strtol0(w[3]) > int(rd["nm"][i]):
                    continue
                s = abs(refio.strtol0(w[1]))
                c = cidx.get(w[0])
                if c is None or c not in has_rows:
                    continue
                if any(tm.rows[int(h)]["name"].upper() != chosen for h in ot.find(c, s, s + qlen)):
                    veto[i] = True
                    break
    return dup, veto
