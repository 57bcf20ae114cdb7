"""-R on the device (iteres_amd/csrc/itx_dedup.hip, C ABI itx_dedup_*) against the reference's loop written down literally
(generic.c:907-919: a key buffer that only a MAPQ >= -Q record refreshes, a hash of the keys seen, `continue` on a hit) —
records handed over window by window like the decoder does, tables that have to grow, keys that share hashes (the overflow
list), streams that begin with MAPQ < -Q records, two files with different reference orders; then the command itself:
`-R` through the device set and through the host's (ITX_HOST_DEDUP=1) must write the same files."""
import filecmp
import os
import subprocess

import numpy as np
import pytest

import refio
from iteres_amd import build, engine as eng, synth

pytestmark = pytest.mark.gpu

CHROM_SIZE = [5_000_000, 3_000_000, 2, 800_000]           # (a size of 2 reads as "not in the size file", generic.c:796-797)
NOLOOKUP = 0x20


def make_records(seed, n, lead_low_mapq=0):
    rng = np.random.default_rng(seed)
    tid = rng.choice([0, 1, 2, 3, 4, -1], n, p=[0.45, 0.3, 0.02, 0.18, 0.03, 0.02]).astype(np.int32)      # 4: beyond the header
    pos = (rng.integers(0, 400, n) * 1000 + rng.integers(0, 12, n)).astype(np.int32)                     # few positions: many equal keys
    pos[rng.random(n) < 0.01] = -3                                                                       # (negative positions exist in the wild)
    length = rng.choice([36, 50, 50, 50, 100, 151, 70_000], n, p=[0.1, 0.3, 0.2, 0.1, 0.2, 0.09, 0.01])
    tmpend = (pos + length).astype(np.int32)
    mapq = rng.choice([0, 3, 9, 10, 20, 37, 60], n).astype(np.uint8)
    if lead_low_mapq:
        mapq[:lead_low_mapq] = 0
    paired = rng.random(n) < 0.35
    f5 = np.zeros(n, np.uint8)
    f5 |= np.where(paired, 1, 0).astype(np.uint8)
    f5 |= np.where(rng.random(n) < 0.03, 2, 0).astype(np.uint8)                    # unmapped
    f5 |= np.where(paired & (rng.random(n) < 0.15), 4, 0).astype(np.uint8)         # mate unmapped
    f5 |= np.where(rng.random(n) < 0.5, 8, 0).astype(np.uint8)                     # reverse
    f5 |= np.where(paired & (rng.random(n) < 0.5), 16, 0).astype(np.uint8)         # read1
    isize = np.where(paired, rng.choice([0, 180, 180, 250, -250, 480, 520, -700], n), 0).astype(np.int32)
    mpos = (pos + np.where(isize < 0, isize + 50, 0)).astype(np.int32)
    return tid, pos, tmpend, mapq, f5, mpos, isize


def model(p, tid2chrom, tid2name, recs, state):
    """the reference's loop, literally; state = [set of keys, key buffer] carried from file to file"""
    tid, pos, tmpend, mapq, f5, mpos, isize = recs
    rd = {"tid": tid, "pos": pos, "tmpend": tmpend, "isize": isize, "mpos": mpos,
          "flag": ((f5 & 1) | ((f5 & 2) << 1) | ((f5 & 4) << 1) | ((f5 & 8) << 1) | ((f5 & 16) << 2)).astype(np.uint16)}
    import goldencase as gc
    drop = np.zeros(len(tid), bool)
    n_uniq_dropped = 0
    seen = state[0]
    for i in range(len(tid)):
        iv = gc.derive_py(p, tid2chrom, CHROM_SIZE, rd, i)
        if iv is None:
            continue
        uniq = int(mapq[i]) >= p["mapq_min"]
        if uniq:
            state[1] = (int(tid2name[tid[i]]), iv[0], iv[1], iv[2])
        if state[1] in seen:
            drop[i] = True
            n_uniq_dropped += uniq
        else:
            seen.add(state[1])
    return drop, n_uniq_dropped


PARAMS = [dict(), dict(extension=0), dict(treat_pe_as_se=True, mapq_min=20), dict(discard_half_mapped=True, isize_max=200, extension=30), dict(mapq_min=0)]


@pytest.mark.parametrize("k", range(len(PARAMS)))
@pytest.mark.parametrize("hash_bits", [None, "17"])
def test_device_set_equals_the_reference_loop(k, hash_bits, monkeypatch):
    import torch
    if hash_bits:
        monkeypatch.setenv("ITX_DEDUP_HASH_BITS", hash_bits)          # 131072 hashes for ~100 k keys: tens of thousands of keys share one with another key, the overflow list settles them
    else:
        monkeypatch.delenv("ITX_DEDUP_HASH_BITS", raising=False)
    p = dict(mapq_min=10, extension=150, isize_max=500, treat_pe_as_se=False, discard_half_mapped=False)
    p.update(PARAMS[k])
    dd = eng.Dedup(CHROM_SIZE, p, first_cells=1 << 16)                # small: the table has to grow several times
    state = [set(), None]
    total_drop = total_uniq = 0
    n = 200_000 if hash_bits else 400_000
    # two "files": the second lists its references in another order (same names -> same ids)
    for fi, (t2c, t2n) in enumerate((([0, 1, 2, 3], [7, 8, 9, 10]), ([3, 0, -1, 1], [10, 7, 11, 8]))):
        recs = make_records(100 + 10 * k + fi, n, lead_low_mapq=5 if fi == 0 else 0)
        dd.set_tidmap(t2c, t2n)
        want, wu = model(p, t2c, t2n, recs, state)
        dev = [torch.from_numpy(a).cuda() for a in recs]
        cuts = [0, 1, 3, 1000, 77_777, n // 2 + 5, n]
        for lo, hi in zip(cuts, cuts[1:]):
            tid, pos, tmpend, mapq, f5, mpos, isize = (a[lo:hi] for a in dev)
            dd.run(tid, pos, tmpend, mapq, f5, mpos, isize)
        got = dev[4].cpu().numpy()
        assert np.array_equal((got & NOLOOKUP) != 0, want), (k, fi, int(((got & NOLOOKUP) != 0).sum()), int(want.sum()))
        assert np.array_equal(got & ~np.uint8(NOLOOKUP), recs[4])     # nothing else was touched
        total_drop += int(want.sum())
        total_uniq += wu
    c = dd.counts()
    assert c["dropped"] == total_drop and c["dup_unique"] == total_uniq, (c, total_drop, total_uniq)
    assert total_drop > n // 4 and c["keys"] > 10_000
    dd.close()


def test_without_mates_and_in_one_piece():
    """unpaired input: no mpos / isize arrays at all; one call for everything"""
    import torch
    recs = list(make_records(5, 100_000))
    recs[4] = (recs[4] & ~np.uint8(1 | 4 | 16)).astype(np.uint8)
    p = dict(mapq_min=10, extension=150, isize_max=500, treat_pe_as_se=False, discard_half_mapped=False)
    want, _ = model(p, [0, 1, 2, 3], [0, 1, 2, 3], recs, [set(), None])
    dd = eng.Dedup(CHROM_SIZE, p)
    dd.set_tidmap([0, 1, 2, 3], [0, 1, 2, 3])
    dev = [torch.from_numpy(a).cuda() for a in recs[:5]]
    dd.run(*dev)
    assert np.array_equal((dev[4].cpu().numpy() & NOLOOKUP) != 0, want)
    dd.close()


# ---- the command ----------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def case(tmp_path_factory):
    d = tmp_path_factory.mktemp("dedup")
    chroms = [("chr1", 40_000_000), ("chr2", 25_000_000), ("chrM", 16_569)]
    t = synth.make_table(31, chroms, 60_000, n_names=400, n_fams=30, n_clas=10, overlap_frac=0.05)
    synth.write_sizes(str(d / "chrom.sizes"), chroms)
    synth.write_sizes(str(d / "rep.sizes"), t.rep_len.items())
    synth.write_rmsk(str(d / "rmsk.txt"), t)
    root = os.path.dirname(build.HERE)
    mk = os.path.join(root, "tools", "mkbam")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-o", mk, os.path.join(root, "tools", "mkbam.c"), "-lz", "-ldl"])
    # pile-ups: thousands of reads at the same positions (exact duplicates by the dozen), mixed CIGARs, and a paired file; XA tags on some
    subprocess.check_call([mk, str(d / "chrom.sizes"), "500000", str(d / "se.bam"), "50", "11", "300", "content=hiseq", "cigar=mixed", "pileup=40"])
    subprocess.check_call([mk, str(d / "chrom.sizes"), "300000", str(d / "pe.bam"), "50", "12", "0", "content=novaseq", "cigar=mixed", "pileup=25", "paired=1"])
    return d


def _run(exe, head, d, aln, out, env):
    os.makedirs(out, exist_ok=True)
    pr = subprocess.run([exe] + head + ["-o", "out", str(d / "chrom.sizes"), str(d / "rep.sizes"), str(d / "rmsk.txt"), aln], cwd=out, capture_output=True,
                        text=True, timeout=600, env=env)
    assert pr.returncode == 0, pr.stderr[-2500:]
    return pr


@pytest.mark.parametrize("head", [["stat", "-w", "-R"], ["stat", "-w", "-R", "-E", "0", "-Q", "30"], ["stat", "-w", "-R", "-B", "-V"], ["filter", "-n", "Rep3", "-R"],
                                  ["filter", "-c", "Cls1", "-R", "-r"]])
def test_command_device_set_equals_host_set(head, case, tmp_path):
    lib, exe = build.build_all()
    d = case
    for aln, what in ((str(d / "se.bam"), "se"), (str(d / "pe.bam"), "pe"), (str(d / "se.bam") + "," + str(d / "pe.bam"), "list")):
        if head[0] == "filter" and what == "list":
            continue
        host_dir, dev_dir = str(tmp_path / f"host_{what}"), str(tmp_path / f"dev_{what}")
        _run(exe, head, d, aln, host_dir, dict(os.environ, ITX_HOST_DEDUP="1", ITX_GPUS="1"))
        pr = _run(exe, head, d, aln, dev_dir, dict(os.environ, ITX_TIMING="1", ITX_GPUS="1", ITX_BGZF_CHUNK="3000000"))      # several windows per file
        assert "-R on the device" in pr.stderr, pr.stderr[-1500:]
        names = sorted(os.listdir(host_dir))
        assert names and names == sorted(os.listdir(dev_dir))
        for fn in names:
            if fn.endswith(".bigWig"):
                assert refio.bigwig_digest(open(os.path.join(host_dir, fn), "rb").read()) == refio.bigwig_digest(open(os.path.join(dev_dir, fn), "rb").read()), fn
            else:
                assert filecmp.cmp(os.path.join(host_dir, fn), os.path.join(dev_dir, fn), shallow=False), (what, fn)
        if what == "se" and head[0] == "stat":
            rep = open(os.path.join(dev_dir, "out.iteres.report")).read()
            assert "non-redundant" in rep or "redundant" in rep.lower()


def test_command_equals_the_reference_binary(case, tmp_path):
    """... and the reference itself (oracle/_ref/iteres travels as a built file): stat -R on the pile-up BAM, every text output"""
    ref = os.path.join(os.path.dirname(build.HERE), "oracle", "_ref", "iteres")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/iteres not built")
    lib, exe = build.build_all()
    d = case
    for aln, what in ((str(d / "se.bam"), "se"), (str(d / "pe.bam"), "pe")):
        a, b = str(tmp_path / f"ref_{what}"), str(tmp_path / f"ours_{what}")
        _run(ref, ["stat", "-w", "-R"], d, aln, a, dict(os.environ))
        _run(exe, ["stat", "-w", "-R"], d, aln, b, dict(os.environ, ITX_GPUS="1"))
        for fn in sorted(os.listdir(a)):
            if not fn.endswith(".bigWig"):
                assert filecmp.cmp(os.path.join(a, fn), os.path.join(b, fn), shallow=False), (what, fn)


def test_device_set_with_windows_the_host_has_to_read(case, tmp_path):
    """A size file that lacks one of the BAM's references: windows with mapped records on it go to the host (the reference warns
    once per such chromosome, in file order) AFTER the device has marked their duplicates — the marks must survive the fetch and
    the host's re-parse of the records' strings (-B / -V); same files as with the host's set, and the warning is there."""
    lib, exe = build.build_all()
    d = case
    short = tmp_path / "chrom_short.sizes"
    short.write_text("".join(ln for ln in open(d / "chrom.sizes") if not ln.startswith("chrM")))

    def run(out, env, opts):
        os.makedirs(out, exist_ok=True)
        pr = subprocess.run([exe, "stat", "-w", "-R"] + opts + ["-o", "out", str(short), str(d / "rep.sizes"), str(d / "rmsk.txt"), str(d / "se.bam")], cwd=out,
                            capture_output=True, text=True, timeout=600, env=env)
        assert pr.returncode == 0, pr.stderr[-2000:]
        return pr
    for opts, what in (([], "plain"), (["-B", "-V"], "bed")):
        host_dir, dev_dir = str(tmp_path / f"host_{what}"), str(tmp_path / f"dev_{what}")
        run(host_dir, dict(os.environ, ITX_HOST_DEDUP="1", ITX_GPUS="1"), opts)
        pr = run(dev_dir, dict(os.environ, ITX_TIMING="1", ITX_GPUS="1", ITX_BGZF_CHUNK="3000000"), opts)
        assert "-R on the device" in pr.stderr and "chrM not existed in the chromosome size file" in pr.stderr, pr.stderr[-1500:]
        for fn in sorted(os.listdir(host_dir)):
            if not fn.endswith(".bigWig"):
                assert filecmp.cmp(os.path.join(host_dir, fn), os.path.join(dev_dir, fn), shallow=False), (what, fn)
