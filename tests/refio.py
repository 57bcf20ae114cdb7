"""Test-side readers for iteres' input and output FILE FORMATS, in plain Python.

Independent of the product's C host code on purpose: the oracle tests go
golden input files -> (this module) -> oracle/liboracle.so -> compare with the
reference's golden output files, so a bug in the product's parsers cannot hide
behind a matching bug here.
"""
from __future__ import annotations

import gzip
import os
import re
import struct
import zlib

import numpy as np


def open_maybe_gz(path, mode="rt"):
    if os.path.exists(path):
        return open(path, mode)
    return gzip.open(path + ".gz", mode)


def read_bytes(path):
    with open_maybe_gz(path, "rb") as f:
        return f.read()


def materialise(src_dir, name, dst_dir):
    """Copy (gunzip if stored gzipped) a golden input file into dst_dir; returns its path."""
    dst = os.path.join(dst_dir, name)
    with open(dst, "wb") as f:
        f.write(read_bytes(os.path.join(src_dir, name)))
    return dst


_NUM = re.compile(r"\s*([+-]?)(0[xX][0-9a-fA-F]+|0[0-7]*|[1-9][0-9]*)")


def strtol0(s: str) -> int:
    """C strtol(s, NULL, 0): optional sign, 0x.. hex, 0.. octal, else decimal; 0 when nothing parses."""
    m = _NUM.match(s)
    if not m:
        return 0
    sign, body = m.groups()
    if body.lower().startswith("0x"):
        v = int(body, 16)
    elif body.startswith("0") and len(body) > 1:
        v = int(body, 8)
    else:
        v = int(body)
    v = -v if sign == "-" else v
    return max(min(v, 2**63 - 1), -2**63)


def u32(v: int) -> int:
    return v & 0xFFFFFFFF


def read_sizes(path):
    """cuskent/obscure.c:139-150 hashNameIntFile: later duplicates shadow earlier ones."""
    out = {}
    with open_maybe_gz(path) as f:
        for line in f:
            if not line.strip() or line.lstrip().startswith("#"):
                continue
            w = line.split()
            out[w[0]] = int(w[1])
    return out


def read_rmsk(path):
    """Rows as the reference parses them (generic.c:1587-1607): whitespace-split, '#' lines skipped."""
    rows = []
    with open_maybe_gz(path) as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            w = line.split()
            strand = w[9][0]
            rows.append({
                "chr": w[5], "start": u32(strtol0(w[6])), "end": u32(strtol0(w[7])), "strand": strand,
                "name": w[10], "cname": w[11], "fname": w[12],
                "cons_start": u32(strtol0(w[13] if strand == "+" else w[15])), "cons_end": u32(strtol0(w[14])),
                "cols": w,
            })
    return rows


FUNMAP = 4
_CIG_ADV = set("MDN")     # samtools 0.1.18 bam_calend: cussamtools/bam.c:17-27


def _cigar_tmpend(pos, ops, l_qseq):
    if not ops:
        return pos + l_qseq
    e = pos
    for op, ln in ops:
        if op in _CIG_ADV:
            e += ln
    return e


def _wrap_i32(v):
    return ((int(v) + 2**31) % 2**32) - 2**31


def read_sam(path):
    """SAM text -> core fields, following samtools-0.1.18's text parser where it matters here:
    a mapped record whose CIGAR is '*' is flagged unmapped (cussamtools/bam_import.c sam_read1)."""
    header, idx = [], {}
    recs = {k: [] for k in ("tid", "pos", "tmpend", "mapq", "flag", "mpos", "isize", "qname", "xa", "nm")}
    with open_maybe_gz(path) as f:
        for line in f:
            if line.startswith("@"):
                if line.startswith("@SQ"):
                    d = dict(x.split(":", 1) for x in line.rstrip("\n").split("\t")[1:])
                    idx[d["SN"]] = len(header)
                    header.append((d["SN"], int(d["LN"])))
                continue
            w = line.rstrip("\n").split("\t")
            flag = int(w[1])
            tid = idx.get(w[2], -1)
            pos = int(w[3]) - 1
            ops = [] if w[5] == "*" else [(m.group(2), int(m.group(1))) for m in re.finditer(r"(\d+)([MIDNSHP=X])", w[5])]
            if not ops and not (flag & FUNMAP):
                flag |= FUNMAP
            l_qseq = 0 if w[9] == "*" else len(w[9])
            recs["tid"].append(tid)
            recs["pos"].append(pos)
            recs["tmpend"].append(_wrap_i32(_cigar_tmpend(pos, ops, l_qseq)))
            recs["mapq"].append(int(w[4]))
            recs["flag"].append(flag)
            recs["mpos"].append(int(w[7]) - 1)
            recs["isize"].append(int(w[8]))
            recs["qname"].append(w[0])
            # the first XA:Z and NM:i among the optional fields (bam_aux_get; bam_aux2i gives 0 without NM)
            xa = next((x[5:] for x in w[11:] if x.startswith("XA:Z:")), None)
            nm = next((int(x[5:]) for x in w[11:] if x.startswith("NM:i:")), 0)
            recs["xa"].append(xa)
            recs["nm"].append(nm if xa is not None else 0)
    return header, _to_arrays(recs)


def _to_arrays(recs):
    return {"tid": np.array(recs["tid"], np.int32), "pos": np.array(recs["pos"], np.int32),
            "tmpend": np.array(recs["tmpend"], np.int32), "mapq": np.array(recs["mapq"], np.uint8),
            "flag": np.array(recs["flag"], np.uint16), "mpos": np.array(recs["mpos"], np.int32),
            "isize": np.array(recs["isize"], np.int32), "qname": recs["qname"], "xa": recs.get("xa"), "nm": recs.get("nm")}


def bgzf_decompress(data: bytes) -> bytes:
    out, off = [], 0
    while off < len(data):
        d = zlib.decompressobj(31)
        out.append(d.decompress(data[off:]))
        off = len(data) - len(d.unused_data)
    return b"".join(out)


def read_bam(path):
    raw = bgzf_decompress(read_bytes(path))
    assert raw[:4] == b"BAM\1"
    l_text, = struct.unpack_from("<i", raw, 4)
    off = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, off)
    off += 4
    header = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, off)
        name = raw[off + 4: off + 4 + l_name - 1].decode()
        ln, = struct.unpack_from("<i", raw, off + 4 + l_name)
        header.append((name, ln))
        off += 8 + l_name
    recs = {k: [] for k in ("tid", "pos", "tmpend", "mapq", "flag", "mpos", "isize", "qname", "xa", "nm")}
    cig_ops = "MIDNSHP=X"
    while off + 4 <= len(raw):
        bs, = struct.unpack_from("<i", raw, off)
        tid, pos, bmn, fnc, l_seq, mtid, mpos, isize = struct.unpack_from("<iiIIiiii", raw, off + 4)
        l_qn = bmn & 0xFF
        mapq = (bmn >> 8) & 0xFF
        flag = fnc >> 16
        n_cig = fnc & 0xFFFF
        p = off + 36
        qname = raw[p: p + l_qn - 1].decode()
        p += l_qn
        ops = []
        for k in range(n_cig):
            c, = struct.unpack_from("<I", raw, p + 4 * k)
            ops.append((cig_ops[c & 0xF] if (c & 0xF) < 9 else "?", c >> 4))
        recs["tid"].append(tid)
        recs["pos"].append(pos)
        recs["tmpend"].append(_wrap_i32(_cigar_tmpend(pos, ops, l_seq)))
        recs["mapq"].append(mapq)
        recs["flag"].append(flag)
        recs["mpos"].append(mpos)
        recs["isize"].append(isize)
        recs["qname"].append(qname)
        aux = _bam_aux(raw[p + 4 * n_cig + (l_seq + 1) // 2 + l_seq: off + 4 + bs])
        recs["xa"].append(aux.get("XA"))
        recs["nm"].append(int(aux.get("NM", 0)) if "XA" in aux else 0)
        off += 4 + bs
    return header, _to_arrays(recs)


def _bam_aux(b: bytes):
    """first value of every aux tag (cussamtools/bam_aux.c:36-48 walk)"""
    out, i = {}, 0
    fixed = {"A": "<c", "c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f", "d": "<d"}
    while i + 3 <= len(b):
        tag, ty = b[i:i + 2].decode(), chr(b[i + 2])
        i += 3
        if ty in fixed:
            v, = struct.unpack_from(fixed[ty], b, i)
            i += struct.calcsize(fixed[ty])
        elif ty in "ZH":
            j = b.index(b"\0", i)
            v = b[i:j].decode()
            i = j + 1
        elif ty == "B":
            sub, cnt = chr(b[i]), struct.unpack_from("<I", b, i + 1)[0]
            i += 5 + cnt * struct.calcsize(fixed[sub])
            v = None
        else:
            break
        out.setdefault(tag, v)
    return out


# ------------------------------------------------------------------------------ outputs of the reference

def parse_stat(path):
    with open_maybe_gz(path) as f:
        lines = f.read().split("\n")
    return lines[0].split("\t"), [l.split("\t") for l in lines[1:] if l]


def parse_wig(path):
    out, cur = {}, None
    order = []
    with open_maybe_gz(path) as f:
        for line in f:
            if line.startswith("fixedStep"):
                name = line.split("chrom=")[1].split(" start=")[0]
                cur = []
                out[name] = cur
                order.append(name)
            elif line.strip():
                cur.append(int(line))
    return {k: np.array(v, np.uint32) for k, v in out.items()}, order


def parse_report(path):
    vals = []
    with open_maybe_gz(path) as f:
        for line in f:
            vals.append(int(line.rstrip("\n").rsplit(": ", 1)[1]))
    # report order (generic.c:55-68): cnt[0], cnt[6], cnt[7], cnt[11], cnt[12], cnt[9], cnt[10]
    return dict(zip((0, 6, 7, 11, 12, 9, 10), vals))


def parse_loci(path):
    with open_maybe_gz(path) as f:
        lines = f.read().split("\n")
    return lines[0].split("\t"), [l.split("\t") for l in lines[1:] if l]


# ------------------------------------------------------------------------------ kent hash iteration order

def kent_hash_string(s: str) -> int:
    """cuskent/hash.c:41-53 (chars are signed on x86-64)."""
    r = 0
    for b in s.encode():
        c = b - 256 if b >= 128 else b
        r = (r + ((r << 3) & 0xFFFFFFFF) + c) & 0xFFFFFFFF
    return r


def kent_hash_order(names_in_insertion_order, power=12):
    """Iteration order of hashFirst/hashNext (cuskent/hash.c:511-552) over a hash built by hashAdd in
    the given order, starting at 2**power buckets and doubling when elCount > size (hash.c:136-140,
    374-410; the resize preserves the newest-first order inside a bucket)."""
    size = 1 << power
    n = 0
    for _ in names_in_insertion_order:
        n += 1
        if n > size:
            size *= 2
    keyed = [((kent_hash_string(nm) & (size - 1)), -i, nm) for i, nm in enumerate(names_in_insertion_order)]
    return [nm for _, _, nm in sorted(keyed)]


# ------------------------------------------------------------------------------ bigWig (decoded content)

def bigwig_decode(data: bytes):
    """Decodes a bigWig into what it says, independent of how its blocks were deflated: chromosomes (walking the
    B+ tree), the data sections and the zoom records (walking the R trees, inflating every block), the zoom
    reductions and the file-wide summary. Format as written by cuskent/bwgCreate.c:887-1019."""
    sig, version, n_zoom, chrom_off, data_off, index_off, fcnt, dfcnt, asql, total_off, unc_buf, _ = struct.unpack_from("<IHHQQQHHQQIQ", data, 0)
    assert sig == 0x888FFC26 and struct.unpack_from("<I", data, len(data) - 4)[0] == sig
    zooms = [struct.unpack_from("<IIQQ", data, 64 + 24 * i) for i in range(n_zoom)]
    total = struct.unpack_from("<Qdddd", data, total_off) if total_off else None

    # chromosome B+ tree (cuskent/bPlusTree.c)
    magic, bsize, ksize, vsize, n_items, _ = struct.unpack_from("<IIIIQQ", data, chrom_off)
    assert magic == 0x78CA8C91 and vsize == 8
    chroms = []

    def walk_bpt(off):
        is_leaf, _, cnt = struct.unpack_from("<BBH", data, off)
        off += 4
        for i in range(cnt):
            key = data[off:off + ksize].split(b"\0", 1)[0].decode()
            if is_leaf:
                cid, size = struct.unpack_from("<II", data, off + ksize)
                chroms.append((key, cid, size))
                off += ksize + 8
            else:
                child, = struct.unpack_from("<Q", data, off + ksize)
                walk_bpt(child)
                off += ksize + 8
    walk_bpt(chrom_off + 32)
    assert len(chroms) == n_items

    def walk_cir(off):
        """leaf items (start chrom, start base, end chrom, end base, offset, size) in tree order"""
        magic, bs, cnt, sc, sb, ec, eb, end_off, per_slot, _ = struct.unpack_from("<IIQIIIIQII", data, off)
        assert magic == 0x2468ACE0
        out = []

        def rec(o):
            is_leaf, _, n = struct.unpack_from("<BBH", data, o)
            o += 4
            for i in range(n):
                if is_leaf:
                    out.append(struct.unpack_from("<IIIIQQ", data, o))
                    o += 32
                else:
                    child, = struct.unpack_from("<Q", data, o + 16)
                    rec(child)
                    o += 24
        rec(off + 48)
        return {"block_size": bs, "item_count": cnt, "bounds": (sc, sb, ec, eb), "items_per_slot": per_slot}, out

    n_sections, = struct.unpack_from("<Q", data, data_off)
    idx_head, leaves = walk_cir(index_off)
    sections = []
    for sc, sb, ec, eb, off, size in leaves:
        raw = zlib.decompress(data[off:off + size])
        cid, start, end, step, span, typ, _, cnt = struct.unpack_from("<IIIIIBBH", raw, 0)
        assert (sc, sb, ec, eb) == (cid, start, cid, end) and typ == 3 and len(raw) == 24 + 4 * cnt
        sections.append((cid, start, end, step, span, typ, raw[24:]))
    assert len(sections) == n_sections == idx_head["item_count"]
    zoom_out = []
    for red, _, zdata, zindex in zooms:
        cnt, = struct.unpack_from("<I", data, zdata)
        zh, zl = walk_cir(zindex)
        recs = b"".join(zlib.decompress(data[off:off + size]) for _, _, _, _, off, size in zl)
        assert len(recs) == 32 * cnt == 32 * zh["item_count"]
        zoom_out.append((red, zh["bounds"], recs))
    return {"version": version, "field_counts": (fcnt, dfcnt, asql), "uncompress_buf": unc_buf, "total": total, "chroms": chroms,
            "index": idx_head, "sections": sections, "zooms": zoom_out}


def bigwig_digest(data: bytes) -> str:
    import hashlib
    d = bigwig_decode(data)
    h = hashlib.sha256()
    h.update(repr((d["version"], d["field_counts"], d["uncompress_buf"], d["total"], d["chroms"], sorted(d["index"].items()))).encode())
    for s in d["sections"]:
        h.update(struct.pack("<IIIIIB", *s[:6]) + s[6])
    for red, bounds, recs in d["zooms"]:
        h.update(struct.pack("<IIIII", red, *bounds) + recs)
    return h.hexdigest()


def bigwig_values(data: bytes):
    """{chromosome name: float32 array} of the per-base values"""
    d = bigwig_decode(data)
    names = {cid: (nm, size) for nm, cid, size in d["chroms"]}
    out = {nm: np.zeros(size, np.float32) for nm, size in names.values()}
    for cid, start, end, step, span, typ, raw in d["sections"]:
        out[names[cid][0]][start:end] = np.frombuffer(raw, "<f4")
    return out
