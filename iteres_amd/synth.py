"""Seeded synthetic inputs for the iteres hot path.

Nothing here is product code: it only manufactures inputs — chromosome-size and
repeat-size files, an rmsk.txt in UCSC's 17-column layout (the columns the
reference reads are listed in /root/reference/generic.c:1594-1607), and
coordinate-sorted alignments as arrays, SAM text or BAM (BGZF) — for the golden
fixture generator, the parity tests and bench.py.
"""
from __future__ import annotations

import struct
import zlib
from dataclasses import dataclass, field

import numpy as np

# flag bits (SAM spec)
FPAIRED, FPROPER, FUNMAP, FMUNMAP, FREVERSE, FMREVERSE, FREAD1, FREAD2 = 1, 2, 4, 8, 16, 32, 64, 128

HG38_CHROMS = [
    ("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555),
    ("chr5", 181538259), ("chr6", 170805979), ("chr7", 159345973), ("chr8", 145138636),
    ("chr9", 138394717), ("chr10", 133797422), ("chr11", 135086622), ("chr12", 133275309),
    ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189), ("chr16", 90338345),
    ("chr17", 83257441), ("chr18", 80373285), ("chr19", 58617616), ("chr20", 64444167),
    ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415),
    ("chrM", 16569),
]


@dataclass
class Table:
    """A synthetic RepeatMasker table, rows in FILE order (= binKeeper insertion order)."""
    chroms: list                      # [(name, size)]
    chrom: np.ndarray                 # int32 [R] index into chroms
    start: np.ndarray                 # int64 [R]
    end: np.ndarray                   # int64 [R]
    strand: np.ndarray                # uint8 [R] ord('+')/ord('-')
    rep_name: np.ndarray              # int32 [R] index into names
    fam_of_row: np.ndarray            # int32 [R] index into fams  (taken from the ROW, not the name)
    cla_of_row: np.ndarray            # int32 [R] index into clas
    cons_start: np.ndarray            # int64 [R] value the reference parses as consensus_start
    cons_end: np.ndarray              # int64 [R]
    names: list = field(default_factory=list)
    fams: list = field(default_factory=list)
    clas: list = field(default_factory=list)
    rep_len: dict = field(default_factory=dict)   # name -> consensus length (names may be missing)
    extra_rmsk_chroms: list = field(default_factory=list)  # chrom names present in rmsk but not in sizes


def make_table(seed: int, chroms, n_intervals: int, n_names: int = 400, n_fams: int = 40, n_clas: int = 12,
               overlap_frac: float = 0.03, shuffle_frac: float = 0.0, missing_len_frac: float = 0.02,
               inconsistent_frac: float = 0.0, median_len: float = 200.0, weird_cons_frac: float = 0.02) -> Table:
    """rmsk-like table: ~50 % of each chromosome covered, log-normal lengths, Zipfian name usage."""
    rng = np.random.default_rng(seed)
    sizes = np.array([s for _, s in chroms], dtype=np.int64)
    share = sizes / sizes.sum()
    per_chrom = np.maximum((share * n_intervals).astype(np.int64), 0)
    per_chrom[np.argmax(per_chrom)] += n_intervals - per_chrom.sum()
    # names / families / classes
    names = [f"Rep{i}" if i % 7 else f"(R{i})n" for i in range(n_names)]
    fams = [f"Fam{i}" for i in range(n_fams)]
    clas = [f"Cls{i}" for i in range(n_clas)]
    fam_of_name = rng.integers(0, n_fams, n_names).astype(np.int32)
    cla_of_fam = rng.integers(0, n_clas, n_fams).astype(np.int32)
    cons_len = np.clip(rng.lognormal(np.log(600.0), 0.9, n_names), 100, 7000).astype(np.int64)
    rep_len = {names[i]: int(cons_len[i]) for i in range(n_names) if rng.random() >= missing_len_frac}
    w = 1.0 / (np.arange(n_names) + 1.0)
    w /= w.sum()
    cols = {k: [] for k in ("chrom", "start", "end")}
    for ci, n in enumerate(per_chrom):
        n = int(n)
        if n == 0:
            continue
        ln = np.clip(rng.lognormal(np.log(median_len), 0.9, n), 10, 6000).astype(np.int64)
        tot = int(ln.sum())
        free = max(int(sizes[ci]) - tot - 2, n)
        gaps = rng.exponential(1.0, n + 1)
        gaps = (gaps / gaps.sum() * free * 0.98).astype(np.int64)
        st = np.cumsum(gaps[:n] + ln) - ln
        if overlap_frac > 0 and n > 2:
            k = rng.random(n) < overlap_frac
            k[0] = False
            back = (rng.random(n) * np.roll(ln, 1) * 1.2).astype(np.int64) + 1
            st = np.where(k, np.maximum(st - np.roll(gaps[:n], 0) - back, 0), st)
        en = np.minimum(st + ln, sizes[ci])
        ok = en > st
        st, en = st[ok], en[ok]
        order = np.argsort(st, kind="stable")
        cols["chrom"].append(np.full(len(st), ci, np.int32))
        cols["start"].append(st[order])
        cols["end"].append(en[order])
    chrom = np.concatenate(cols["chrom"])
    start = np.concatenate(cols["start"])
    end = np.concatenate(cols["end"])
    R = len(start)
    rep = rng.choice(n_names, size=R, p=w).astype(np.int32)
    fam = fam_of_name[rep].copy()
    if inconsistent_frac > 0:   # same repName seen with another family string on some rows
        k = rng.random(R) < inconsistent_frac
        fam[k] = rng.integers(0, n_fams, int(k.sum()))
    cla = cla_of_fam[fam].copy()
    if inconsistent_frac > 0:
        k = rng.random(R) < inconsistent_frac
        cla[k] = rng.integers(0, n_clas, int(k.sum()))
    strand = np.where(rng.random(R) < 0.5, ord("+"), ord("-")).astype(np.uint8)
    L = cons_len[rep]
    glen = end - start
    cs = (rng.random(R) * np.maximum(L - glen, 1)).astype(np.int64)
    ce = cs + glen + (rng.normal(0, 0.05, R) * glen).astype(np.int64)
    ce = np.maximum(ce, cs + 1)
    if weird_cons_frac > 0:
        k = rng.random(R) < weird_cons_frac
        ce = np.where(k, L + rng.integers(1, 50, R), ce)          # runs past the consensus: j >= length break
        k2 = rng.random(R) < weird_cons_frac / 4
        ce = np.where(k2, np.maximum(cs - 3, 0), ce)              # consensus_end <= consensus_start
    if shuffle_frac > 0:   # file order != coordinate order for a slice of rows
        idx = np.arange(R)
        k = np.flatnonzero(rng.random(R) < shuffle_frac)
        idx[k] = rng.permutation(k)
        chrom, start, end, strand, rep, fam, cla, cs, ce = (a[idx] for a in (chrom, start, end, strand, rep, fam, cla, cs, ce))
    return Table(list(chroms), chrom.astype(np.int32), start, end, strand, rep, fam.astype(np.int32),
                 cla.astype(np.int32), cs, ce, names, fams, clas, rep_len)


def write_sizes(path, pairs):
    with open(path, "w") as f:
        for n, s in pairs:
            f.write(f"{n}\t{s}\n")


def _rmsk_lines(t: Table, lo: int, hi: int) -> str:
    out = []
    for i in range(lo, hi):
        nm = t.names[t.rep_name[i]]
        L = t.rep_len.get(nm, 0)
        cs, ce = int(t.cons_start[i]), int(t.cons_end[i])
        left = -(max(L - ce, 0))
        c13, c15 = (cs, left) if t.strand[i] == ord("+") else (left, cs)
        cname, csize = t.chroms[t.chrom[i]]
        out.append(f"{585 + (int(t.start[i]) >> 17)}\t{1000 + i % 977}\t{i % 300}\t{i % 40}\t{i % 30}\t{cname}\t"
                   f"{int(t.start[i])}\t{int(t.end[i])}\t{-(csize - int(t.end[i]))}\t{chr(t.strand[i])}\t{nm}\t"
                   f"{t.clas[t.cla_of_row[i]]}\t{t.fams[t.fam_of_row[i]]}\t{c13}\t{ce}\t{c15}\t{i % 9 + 1}\n")
    return "".join(out)


_RMSK_JOB = None


def _rmsk_chunk(b):
    return _rmsk_lines(_RMSK_JOB, b[0], b[1])


def write_rmsk(path, t: Table, extra_rows=(), workers: int = 1):
    """17 UCSC columns: bin swScore milliDiv milliDel milliIns genoName genoStart genoEnd genoLeft
    strand repName repClass repFamily repStart repEnd repLeft id. workers > 1: pieces of the table are formatted by
    forked helper processes (the text is the same)."""
    global _RMSK_JOB
    n = len(t.start)
    with open(path, "w") as f:
        if workers > 1 and n > 200_000:
            import multiprocessing as mp
            step = 100_000
            _RMSK_JOB = t
            try:
                with mp.get_context("fork").Pool(workers) as pool:
                    for txt in pool.imap(_rmsk_chunk, [(a, min(a + step, n)) for a in range(0, n, step)]):
                        f.write(txt)
            finally:
                _RMSK_JOB = None
        else:
            f.write(_rmsk_lines(t, 0, n))
        for row in extra_rows:
            f.write("\t".join(str(x) for x in row) + "\n")


@dataclass
class Reads:
    """Alignment records as BAM core fields, in file order."""
    header: list                 # [(name, len)] @SQ lines == BAM reference list (tid order)
    tid: np.ndarray              # int32
    pos: np.ndarray              # int32 0-based
    flag: np.ndarray             # uint16
    mapq: np.ndarray             # uint8
    l_qseq: np.ndarray           # int32
    mtid: np.ndarray             # int32
    mpos: np.ndarray             # int32
    isize: np.ndarray            # int32
    cigars: list                 # list[list[(op_char, len)]] ([] = '*')
    qname: list
    aux: list | None = None      # optional list[bytes-like str] of SAM aux text fields e.g. ["NM:i:1", "XA:Z:..."]

    def __len__(self):
        return len(self.tid)

    def calend(self):
        """samtools-0.1.18 bam_calend (cussamtools/bam.c:17-27): only M, D, N advance; '=' and 'X' do not."""
        out = np.empty(len(self), np.int64)
        for i, cg in enumerate(self.cigars):
            e = int(self.pos[i])
            for op, ln in cg:
                if op in "MDN":
                    e += ln
            out[i] = e
        return out

    def tmpend(self):
        """generic.c:820 — n_cigar ? bam_calend : pos + l_qseq, as the int32 the reference stores."""
        ce = self.calend()
        nc = np.array([len(c) for c in self.cigars])
        te = np.where(nc > 0, ce, self.pos.astype(np.int64) + self.l_qseq)
        return te.astype(np.int64).astype(np.int32)


def make_reads(seed: int, header, n: int, read_len=(36, 100), paired_frac: float = 0.0, unmapped_frac: float = 0.02,
               odd_cigar_frac: float = 0.08, sorted_: bool = True, tids=None, nocigar_frac: float = 0.01) -> Reads:
    rng = np.random.default_rng(seed)
    sizes = np.array([s for _, s in header], dtype=np.int64)
    use = np.arange(len(header)) if tids is None else np.asarray(tids)
    p = sizes[use] / sizes[use].sum()
    tid = use[rng.choice(len(use), size=n, p=p)].astype(np.int32)
    rl = rng.integers(read_len[0], read_len[1] + 1, n).astype(np.int32)
    pos = (rng.random(n) * np.maximum(sizes[tid] - 1, 1)).astype(np.int64)   # may run past the end: end is clipped
    flag = np.zeros(n, np.uint16)
    flag |= np.where(rng.random(n) < 0.5, FREVERSE, 0).astype(np.uint16)
    mapq = rng.choice(np.array([0, 0, 3, 20, 37, 37, 37, 60], np.uint8), n)
    mtid = np.full(n, -1, np.int32)
    mpos = np.full(n, -1, np.int32)
    isize = np.zeros(n, np.int32)
    if paired_frac > 0:
        pe = rng.random(n) < paired_frac
        k = np.flatnonzero(pe)
        flag[k] |= FPAIRED
        r1 = rng.random(len(k)) < 0.5
        flag[k] |= np.where(r1, FREAD1, FREAD2).astype(np.uint16)
        mun = rng.random(len(k)) < 0.1
        flag[k] |= np.where(mun, FMUNMAP, 0).astype(np.uint16)
        ins = np.clip(rng.normal(350, 90, len(k)), 0, 900).astype(np.int32)
        ins[rng.random(len(k)) < 0.03] = 0
        sign = np.where(rng.random(len(k)) < 0.5, 1, -1)
        isize[k] = np.where(mun, 0, ins * sign)
        mtid[k] = np.where(mun, -1, tid[k])
        mpos[k] = np.where(mun, -1, np.maximum(pos[k] - np.where(sign < 0, ins - rl[k], 0), 0))
    un = rng.random(n) < unmapped_frac
    flag[un] |= FUNMAP
    if sorted_:
        order = np.lexsort((pos, tid))
        tid, pos, flag, mapq, rl, mtid, mpos, isize = (a[order] for a in (tid, pos, flag, mapq, rl, mtid, mpos, isize))
    cigars = []
    odd = rng.random(n) < odd_cigar_frac
    noc = rng.random(n) < nocigar_frac
    for i in range(n):
        L = int(rl[i])
        if noc[i]:
            cigars.append([])
        elif not odd[i] or L < 20:
            cigars.append([("M", L)])
        else:
            kind = rng.integers(0, 6)
            a = int(rng.integers(5, L - 10))
            if kind == 0:
                cigars.append([("M", a), ("D", int(rng.integers(1, 8))), ("M", L - a)])
            elif kind == 1:
                cigars.append([("M", a), ("I", 2), ("M", L - a - 2)])
            elif kind == 2:
                cigars.append([("S", 4), ("M", L - 4)])
            elif kind == 3:
                cigars.append([("M", a), ("N", int(rng.integers(50, 3000))), ("M", L - a)])
            elif kind == 4:
                cigars.append([("=", a), ("X", 1), ("M", L - a - 1)])     # '=' / 'X' do not advance calend in 0.1.18
            else:
                cigars.append([("H", 3), ("M", L), ("P", 1)])
    qname = [f"r{i}" for i in range(n)]
    return Reads(list(header), tid, pos.astype(np.int32), flag, mapq.astype(np.uint8), rl, mtid, mpos, isize, cigars, qname)


_SEQ = "ACGT"


def _seq_for(i, L):
    return "".join(_SEQ[(i * 7 + j * 3 + (j >> 2)) & 3] for j in range(L))


def write_sam(path, r: Reads, with_seq: bool = True):
    with open(path, "w") as f:
        f.write("@HD\tVN:1.0\tSO:coordinate\n")
        for nme, ln in r.header:
            f.write(f"@SQ\tSN:{nme}\tLN:{ln}\n")
        for i in range(len(r)):
            tid = int(r.tid[i])
            rname = r.header[tid][0] if tid >= 0 else "*"
            cg = "".join(f"{ln}{op}" for op, ln in r.cigars[i]) or "*"
            mt = int(r.mtid[i])
            rnext = "*" if mt < 0 else ("=" if mt == tid else r.header[mt][0])
            L = int(r.l_qseq[i])
            seq = _seq_for(i, L) if (with_seq and L > 0) else "*"
            qual = "I" * L if (with_seq and L > 0) else "*"
            aux = ("\t" + "\t".join(r.aux[i])) if (r.aux is not None and r.aux[i]) else ""
            f.write(f"{r.qname[i]}\t{int(r.flag[i])}\t{rname}\t{int(r.pos[i]) + 1}\t{int(r.mapq[i])}\t{cg}\t{rnext}\t"
                    f"{int(r.mpos[i]) + 1}\t{int(r.isize[i])}\t{seq}\t{qual}{aux}\n")


_CIG = {c: i for i, c in enumerate("MIDNSHP=X")}
_NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def _reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def _aux_bin(fields):
    out = bytearray()
    for fld in fields or ():
        tag, ty, val = fld.split(":", 2)
        out += tag.encode()
        if ty == "i":
            v = int(val)
            if 0 <= v < 256:
                out += b"C" + struct.pack("<B", v)
            elif -32768 <= v < 32768:
                out += b"s" + struct.pack("<h", v)
            else:
                out += b"i" + struct.pack("<i", v)
        elif ty == "Z":
            out += b"Z" + val.encode() + b"\0"
        elif ty == "A":
            out += b"A" + val.encode()[:1]
        elif ty == "B":                                  # "XY:B:<hex>": a uint8 array holding exactly these bytes
            raw = bytes.fromhex(val)
            out += b"BC" + struct.pack("<I", len(raw)) + raw
        else:
            raise ValueError(ty)
    return bytes(out)


def bgzf_block(data: bytes, level: int = 1) -> bytes:
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize)
            + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def write_bam(path, r: Reads, with_seq: bool = True, level: int = 1, eof: bool = True, block: int = 0xff00):
    text = "@HD\tVN:1.0\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in r.header)
    buf = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(r.header)))
    for n, l in r.header:
        nb = n.encode() + b"\0"
        buf += struct.pack("<i", len(nb)) + nb + struct.pack("<i", l)
    ce = r.calend()
    with open(path, "wb") as f:
        def flush(final=False):
            nonlocal buf
            while len(buf) >= block or (final and len(buf) > 0):
                f.write(bgzf_block(bytes(buf[:block]), level))
                del buf[:block]
        for i in range(len(r)):
            qn = r.qname[i].encode() + b"\0"
            cg = r.cigars[i]
            L = int(r.l_qseq[i]) if with_seq else 0
            pos = int(r.pos[i])
            endp = int(ce[i]) if cg else pos + 1
            b = _reg2bin(max(pos, 0), max(endp, pos + 1))
            body = struct.pack("<iiIIiiii", int(r.tid[i]), pos, (b << 16) | (int(r.mapq[i]) << 8) | len(qn),
                               (int(r.flag[i]) << 16) | len(cg), L, int(r.mtid[i]), int(r.mpos[i]), int(r.isize[i]))
            body += qn + b"".join(struct.pack("<I", (ln << 4) | _CIG[op]) for op, ln in cg)
            if L:
                s = _seq_for(i, L)
                nib = [_NT16[c] for c in s] + [0]
                body += bytes((nib[2 * j] << 4) | nib[2 * j + 1] for j in range((L + 1) // 2)) + bytes([40]) * L
            body += _aux_bin(r.aux[i] if r.aux is not None else None)
            buf += struct.pack("<i", len(body)) + body
            flush()
        flush(final=True)
        if eof:
            f.write(BGZF_EOF)


# ---------------------------------------------------------------- large, array-only workloads (bench / full-size tests)

def make_reads_soa(seed: int, header, n: int, read_len=(100, 150), odd_cigar_frac: float = 0.05, chunk: int = 1 << 24):
    """Coordinate-sorted single-end alignments straight to the engine's record SoA
    (tid i32, pos i32, tmpend i32, mapq u8, flag5 u8) without materialising text: SURVEY §8(d) cfg2/cfg3 shape."""
    rng = np.random.default_rng(seed)
    sizes = np.array([s for _, s in header], dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(sizes)])
    g = np.sort(rng.integers(0, cum[-1], n, dtype=np.int64))
    tid = (np.searchsorted(cum, g, side="right") - 1).astype(np.int32)
    pos = (g - cum[tid]).astype(np.int32)
    del g
    rl = rng.integers(read_len[0], read_len[1] + 1, n, dtype=np.int32)
    odd = rng.random(n) < odd_cigar_frac
    extra = np.where(odd, rng.integers(-4, 400, n, dtype=np.int32), 0).astype(np.int32)   # D/N lengthen, I/S shorten
    tmpend = (pos + rl + extra).astype(np.int32)
    mapq = rng.choice(np.array([0, 0, 3, 20, 37, 37, 37, 60], np.uint8), n)
    flag5 = np.where(rng.random(n) < 0.5, 8, 0).astype(np.uint8)      # bit3 = reverse strand
    return tid, pos, tmpend, mapq, flag5


def make_reads_device(seed: int, header, n: int, device):
    """make_reads_soa's distribution made where it is used — in HBM, with torch's device generator (bench.py's resident
    replay and the BASELINE-size GPU tests: 500 M records are 7 GB): uniform sorted positions over the genome, 100-150 bp,
    5 % with a CIGAR that moves the end, MAPQ from the same multiset, half reverse. Returns a dict of contiguous device
    tensors tid/pos/tmpend (int32) and mapq/flag5 (uint8). Not the same draws as the numpy generator."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    sizes = np.array([s for _, s in header], dtype=np.int64)
    cum = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int64, device=device)
    gpos = torch.sort(torch.randint(0, int(cum[-1]), (n,), generator=g, device=device, dtype=torch.int64)).values
    tid64 = torch.bucketize(gpos, cum, right=True) - 1
    pos = (gpos - cum[tid64]).to(torch.int32)
    tid = tid64.to(torch.int32)
    del gpos, tid64
    rl = torch.randint(100, 151, (n,), generator=g, device=device, dtype=torch.int32)
    odd = torch.rand(n, generator=g, device=device) < 0.05
    extra = torch.where(odd, torch.randint(-4, 400, (n,), generator=g, device=device, dtype=torch.int32), torch.zeros((), dtype=torch.int32, device=device))
    tmpend = (pos + rl + extra).contiguous()
    del rl, odd, extra
    mq = torch.tensor([0, 0, 3, 20, 37, 37, 37, 60], dtype=torch.uint8, device=device)
    mapq = mq[torch.randint(0, 8, (n,), generator=g, device=device)].contiguous()
    f5 = torch.where(torch.rand(n, generator=g, device=device) < 0.5, 8, 0).to(torch.uint8).contiguous()
    return {"tid": tid.contiguous(), "pos": pos.contiguous(), "tmpend": tmpend, "mapq": mapq, "flag5": f5}
