"""The XA / NM veto on the device (iteres_amd/csrc/itx_xaveto.hip: generic.c:303-341, 972-982 for windows that stay in
HBM) against the host's reading of the same strings (ITX_HOST_VETO=1: iteres_amd/host/side.c, the route every window took
before) and against the reference binary, on tags written to probe the corners: empty alternatives, no trailing ';', more
than 100 alternatives (chopByChar stops at 100), NM of every integer type and absent, NM' above / equal / below NM, names that
are prefixes of each other or unknown, positions at the ends of a chromosome, repeat names that differ only in case (sameWord),
XA tags that are not strings — and numbers that strtol(.., 0, 0) reads differently from plain decimal (leading zeros, 0x,
blanks), for which the device must step back and let the host decide."""
import filecmp
import os
import struct
import subprocess

import numpy as np
import pytest

from iteres_amd import build, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "iteres")


@pytest.fixture(scope="module")
def exe():
    lib, exe = build.build_all()
    return exe


def _case(tmp, seed, n_reads, weird, ref_safe=True):
    chroms = [("chr1", 3_000_000), ("chr10", 900_000), ("chr1_alt", 400_000), ("chrEmpty", 50_000)]
    t = synth.make_table(seed, chroms[:3], 6000, n_names=90, n_fams=15, n_clas=6, overlap_frac=0.06, shuffle_frac=0.03)
    # names that differ only in case are ONE subfamily for the veto (sameWord)
    t.names = [nm if i % 9 else nm.upper() for i, nm in enumerate(t.names)]
    for i in range(0, len(t.names) - 1, 7):
        t.names[i + 1] = t.names[i].lower() if t.names[i].upper() != t.names[i].lower() else t.names[i + 1]
    t.rep_len = {nm: 400 + 13 * i for i, nm in enumerate(t.names)}
    synth.write_sizes(str(tmp / "chrom.sizes"), chroms)
    synth.write_sizes(str(tmp / "rep.sizes"), t.rep_len.items())
    synth.write_rmsk(str(tmp / "rmsk.txt"), t)
    header = chroms + [("chrZ", 20_000)]
    r = synth.make_reads(seed + 1, header, n_reads, read_len=(30, 120), paired_frac=0.15, odd_cigar_frac=0.05)
    rng = np.random.default_rng(seed + 2)
    names = [nm for nm, _ in chroms]
    aux = [[] for _ in range(len(r))]

    def alt(kind=None):
        kind = rng.random() if kind is None else kind
        if kind < 0.6:
            row = int(rng.integers(0, len(t.start)))
            c, p0 = names[int(t.chrom[row])], int(t.start[row]) + int(rng.integers(-30, 60))
        elif kind < 0.75:
            ci = int(rng.integers(0, 4))
            c, p0 = names[ci], int(rng.choice([1, 2, chroms[ci][1] - 5, chroms[ci][1], chroms[ci][1] + 40, int(rng.integers(1, chroms[ci][1]))]))
        elif kind < 0.85:
            c, p0 = str(rng.choice(["chr", "chr100", "chr1_al", "chr1_alt2", "CHR1", ""])), int(rng.integers(1, 5000))
        else:
            c, p0 = "chrZ", int(rng.integers(1, 20000))
        return f"{c},{'+' if rng.random() < 0.5 else '-'}{max(p0, 0)},{int(rng.integers(20, 120))}M,{int(rng.integers(0, 6))}"

    for i in np.flatnonzero(rng.random(len(r)) < 0.5):
        style = rng.random()
        if style < 0.55:
            s = ";".join(alt() for _ in range(int(rng.integers(1, 6)))) + ";"
        elif style < 0.65:
            s = ";".join(alt() for _ in range(int(rng.integers(1, 4))))              # no trailing ';'
        elif style < 0.72:
            s = ";;" + alt() + ";;" + alt() + ";"                                    # empty alternatives
        elif style < 0.76:
            s = ""                                                                   # an empty tag
        elif style < 0.80:
            # > 100 alternatives: only the first 100 count; the ones behind would veto on almost any row
            quiet = f"chrZ,+5,30M,0"
            s = ";".join([quiet] * 100 + [alt(0.1) for _ in range(12)]) + ";"
        elif style < 0.86:
            s = alt() + ",extra,fields;" + alt() + ";"                               # more than four fields: the fourth ends at the comma
        elif style < 0.90 and weird:
            c = names[int(rng.integers(0, 3))]
            s = str(rng.choice([f"{c},+0644,36M,1;", f"{c},-0x1f40,36M,0;", f"{c}, 812,36M,0;", f"{c},+1234567890123,36M,0;", f"{c},+700,36M,010;"])) + alt() + ";"
        else:
            s = alt(0.05) + ";"
        fields = [f"XA:Z:{s}"]
        k = rng.random()
        if k < 0.85:
            fields.insert(0, f"NM:i:{int(rng.choice([0, 1, 2, 3, 5, 200, 300, 70000, -1]))}")     # C, s, i encodings; a negative one
        if rng.random() < 0.3:
            fields.insert(0, "X0:i:1")
        if rng.random() < 0.2:
            fields.append("ZZ:B:00112233")
        aux[i] = fields
    # a few XA tags that are not strings: bam_aux2Z gives the reference nothing to copy and it crashes there (strcpy of NULL);
    # the drop-in reads them as empty on both routes
    if not ref_safe:
        for i in rng.choice(len(r), 40, replace=False):
            aux[i] = ["NM:i:1", "XA:i:7"]
    r.aux = aux
    synth.write_bam(str(tmp / "reads.bam"), r, with_seq=True)
    return tmp


def _run(exe, tmp, out, env=None, opts=("-w",)):
    os.makedirs(out, exist_ok=True)
    pr = subprocess.run([exe, "stat"] + list(opts) + ["-o", "out", str(tmp / "chrom.sizes"), str(tmp / "rep.sizes"), str(tmp / "rmsk.txt"), str(tmp / "reads.bam")], cwd=out,
                        capture_output=True, text=True, timeout=600, env=dict(os.environ, ITX_TIMING="1", **(env or {})))
    assert pr.returncode == 0, pr.stderr[-2000:]
    return pr


def _same(a, b):
    names = sorted(fn for fn in os.listdir(a) if not fn.endswith(".bigWig"))
    assert len(names) == 6
    for fn in names:
        assert filecmp.cmp(os.path.join(a, fn), os.path.join(b, fn), shallow=False), fn


def _routes(err):
    """(batches judged on the device, batches judged by the host) from the ITX_TIMING line"""
    import re
    m = re.search(r"XA veto: (\d+) batches judged on the device, (\d+) by the host", err)
    assert m, err[-1500:]
    return int(m.group(1)), int(m.group(2))


def _vetoed(d):
    return int(open(os.path.join(d, "out.iteres.report")).read().split("\n")[4].rsplit(":", 1)[1])


@pytest.mark.parametrize("opts", [("-w",), ("-w", "-T", "-E", "0"), ("-w", "-Q", "30", "-I", "800")])
def test_device_veto_equals_host_and_reference(opts, exe, tmp_path):
    """Plain-decimal tags only: every window stays in HBM, the device reads the tags; same files as the host's reading and as
    the reference binary's, and the veto did fire."""
    d = _case(tmp_path, 600 + len(opts), 40_000, weird=False)
    dev = _run(exe, d, str(tmp_path / "dev"), opts=opts, env={"ITX_BGZF_CHUNK": "200000"})
    host = _run(exe, d, str(tmp_path / "host"), opts=opts, env={"ITX_HOST_VETO": "1", "ITX_BGZF_CHUNK": "200000"})
    _same(str(tmp_path / "dev"), str(tmp_path / "host"))
    assert _vetoed(str(tmp_path / "dev")) > 500
    # the device route was taken (no records fetched for the host's passes), the host route was not
    assert _routes(dev.stderr)[0] > 0 and _routes(host.stderr)[0] == 0
    if os.path.exists(REF):
        _run(REF, d, str(tmp_path / "ref"), opts=opts)
        _same(str(tmp_path / "dev"), str(tmp_path / "ref"))


def test_xa_tags_that_are_not_strings(exe, tmp_path):
    """XA:i:7 — no string to chop: nothing to veto, on the device as on the host (the reference crashes on such a file)."""
    d = _case(tmp_path, 650, 20_000, weird=False, ref_safe=False)
    _run(exe, d, str(tmp_path / "dev"))
    _run(exe, d, str(tmp_path / "host"), env={"ITX_HOST_VETO": "1"})
    _same(str(tmp_path / "dev"), str(tmp_path / "host"))


def test_numbers_only_strtol_can_read_go_to_the_host(exe, tmp_path):
    """Leading zeros (octal for strtol base 0), 0x.., a leading blank, thirteen digits: the device declines such a window
    and the host reads it — the files are the reference's all the same."""
    d = _case(tmp_path, 700, 30_000, weird=True)
    dev = _run(exe, d, str(tmp_path / "dev"), env={"ITX_BGZF_CHUNK": "150000"})
    assert _routes(dev.stderr)[1] > 0                               # some windows did go to the host
    _run(exe, d, str(tmp_path / "host"), env={"ITX_HOST_VETO": "1"})
    _same(str(tmp_path / "dev"), str(tmp_path / "host"))
    if os.path.exists(REF):
        _run(REF, d, str(tmp_path / "ref"))
        _same(str(tmp_path / "dev"), str(tmp_path / "ref"))


def test_windows_of_several_batches_stay_on_the_device(exe, tmp_path):
    """A BAM whose records carry no sequence: one decoded window holds more records than a batch of the engine (4 M). The
    veto runs batch by batch inside the window (it used to send such windows to the host, silently 3.7x slower): 9 M records,
    XA tags on a share of them; same files as the host's reading of the tags, and no batch was judged by the host."""
    chroms = [("chr1", 60_000_000), ("chr2", 35_000_000)]
    t = synth.make_table(88, chroms, 70_000, n_names=300, n_fams=25, n_clas=8, overlap_frac=0.05)
    synth.write_sizes(str(tmp_path / "chrom.sizes"), chroms)
    synth.write_sizes(str(tmp_path / "rep.sizes"), t.rep_len.items())
    synth.write_rmsk(str(tmp_path / "rmsk.txt"), t)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mk = os.path.join(root, "tools", "mkbam")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-o", mk, os.path.join(root, "tools", "mkbam.c"), "-lz", "-ldl"])
    subprocess.check_call([mk, str(tmp_path / "chrom.sizes"), "9000000", str(tmp_path / "reads.bam"), "0", "21", "250"])      # no SEQ / QUAL; XA on 25 %
    dev = _run(exe, tmp_path, str(tmp_path / "dev"))
    host = _run(exe, tmp_path, str(tmp_path / "host"), env={"ITX_HOST_VETO": "1"})
    _same(str(tmp_path / "dev"), str(tmp_path / "host"))
    n_dev, n_host = _routes(dev.stderr)
    assert n_dev >= 3 and n_host == 0, (n_dev, n_host)                # 9 M records in one or two windows: at least three batches, all on the device
    assert _vetoed(str(tmp_path / "dev")) > 1000
