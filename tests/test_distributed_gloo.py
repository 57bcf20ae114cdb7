"""The N > 1 path on CPU (gloo, world size 2) with the PRODUCT's share arithmetic: the compressed bytes of the alignment
files are cut into per-rank ranges by iteres_amd/host/shares.c (plan_shares) and turned into record boundaries by
iteres_amd/host/bamio.c (find_split) — both reached through the CPU-only tool iteres_amd/host/test/share_dump.c. Every rank
decodes what lies between ITS two split points (an independent Python walk of the BGZF blocks), produces its partial with
the oracle (no GPU in this container), ONE sum-all-reduce merges the partials (unsigned sums carried in signed tensors:
mod 2^64 / 2^32 like the reference's counters and like ncclSum on the device, csrc/itx_comm.hip) and rank 0 compares
with the single-process result. What is under test: the shares cover every record exactly once, whatever the world size
and the number of files, and the exchange's arithmetic."""
import os
import socket
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HOST = os.path.join(ROOT, "iteres_amd", "host")
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from iteres_amd import engine as eng, synth  # noqa: E402

SIZE_MAX = (1 << 64) - 1
CHROMS = [("c1", 6_000_000), ("c2", 2_000_000)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def build_share_dump(dst_dir):
    exe = os.path.join(dst_dir, "share_dump")
    subprocess.check_call(["gcc", "-O2", "-g", "-fopenmp", "-std=gnu11", "-o", exe, os.path.join(HOST, "test", "share_dump.c"), os.path.join(HOST, "shares.c"),
                           os.path.join(HOST, "bamio.c"), os.path.join(HOST, "tables.c"), "-lz", "-ldl"])
    return exe


@pytest.fixture(scope="module")
def share_dump(tmp_path_factory):
    return build_share_dump(str(tmp_path_factory.mktemp("bin")))


def plan(exe, files, world, min_share=1):
    """{(rank, file index): dict(lo, hi, lo_split, hi_split)} and whether the job is shared at all"""
    out = subprocess.run([exe, str(world), str(min_share), ",".join(files)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().split("\n")
    shared = int(lines[0].split()[1])
    res = {}
    for ln in lines[1:]:
        f = ln.replace("|", " ").split()
        rank, fi, lo, hi = int(f[0]), int(f[1]), int(f[2]), int(f[3])
        res[(rank, fi)] = {"lo": lo, "hi": hi, "lo_split": tuple(int(x) for x in f[4:7]), "hi_split": tuple(int(x) for x in f[7:10])}
    return shared, res


def walk(path):
    """Independent of the product: (inflated stream, {block file offset: inflated offset}, record starts in the stream)"""
    comp = open(path, "rb").read()
    blocks = eng.index_bgzf(comp)
    u = b"".join(zlib.decompress(comp[int(b["coff"]) + 18:int(b["coff"]) + int(b["csize"]) - 8], -15) for b in blocks)
    uoff = {int(b["coff"]): int(b["uoff"]) for b in blocks}
    p = 4
    p += 4 + struct.unpack_from("<i", u, p)[0]
    nref = struct.unpack_from("<i", u, p)[0]
    p += 4
    for _ in range(nref):
        p += 4 + struct.unpack_from("<i", u, p)[0] + 4
    starts = []
    while p + 4 <= len(u):
        starts.append(p)
        p += 4 + struct.unpack_from("<i", u, p)[0]
    return u, uoff, starts


def share_records(sh, uoff, starts, n_stream):
    """indices into `starts` of the records a share holds: from its lo split point up to its hi split point"""
    if sh["lo"] == sh["hi"]:
        return 0, 0
    f0, b0, o0 = sh["lo_split"]
    f1, b1, o1 = sh["hi_split"]
    assert f0 != -1 and f1 != -1, "a split search was given up"
    first = 0 if f0 == 2 else (len(starts) if f0 == 0 else None)
    if first is None:
        first = starts.index(uoff[b0] + o0)                      # (raises when the point is no record start)
    last = len(starts) if f1 in (0, 2) else starts.index(uoff[b1] + o1)
    return first, max(first, last)


def soa_of(u, starts, lo, hi):
    """tid, pos, tmpend, mapq, flag of records [lo, hi) (SE reads with plain nM CIGARs or none: end = pos + l_seq)"""
    n = hi - lo
    tid, pos, tmpend = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.int32)
    mapq, flag = np.empty(n, np.uint8), np.empty(n, np.uint16)
    for k in range(n):
        p = starts[lo + k]
        _, t, ps, bmq, fn, lseq = struct.unpack_from("<iiiIIi", u, p)
        l_name, mq, n_cig, fl = bmq & 0xff, (bmq >> 8) & 0xff, fn & 0xffff, fn >> 16
        end = ps + lseq
        if n_cig:
            end = ps
            for c in struct.unpack_from(f"<{n_cig}I", u, p + 36 + l_name):
                if (c & 15) in (0, 2, 3):
                    end += c >> 4
        tid[k], pos[k], tmpend[k], mapq[k], flag[k] = t, ps, end, mq, fl
    return tid, pos, tmpend, mapq, flag


def _table():
    t = synth.make_table(5, CHROMS, 9000, n_names=120, n_fams=15, n_clas=6, overlap_frac=0.05, inconsistent_frac=0.02)
    rl = np.array([t.rep_len.get(n, 0) for n in t.names], np.uint32)
    rows = eng.make_rows(t.chrom, t.start, t.end, t.cons_start, t.cons_end, t.rep_name, t.fam_of_row, t.cla_of_row)
    return t, rl, rows


def _oracle(t, rl, rows, soa):
    from oracle import binding as orc
    ot = orc.OracleTable([s for _, s in CHROMS], rl, len(t.fams), len(t.clas))
    ot.add_rows(rows["chrom"], rows["start"], rows["end"], rows["cons_start"], rows["cons_end"], rows["rep"], rows["fam"], rows["cla"])
    r = ot.run({}, [0, 1], *soa, want_hits=False)
    ot.close()
    return r


def _signed(a):
    return a.view({np.dtype("uint64"): np.int64, np.dtype("uint32"): np.int32}[a.dtype])


KEYS = ("cnt", "rep_cnt", "fam_cnt", "cla_cnt", "cov", "cov_uniq")


def _rank(rank, world, port, q, exe, files):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shared, pl = plan(exe, files, world)
    assert shared == 1
    t, rl, rows = _table()
    parts = []
    n_mine = 0
    for fi, path in enumerate(files):
        u, uoff, starts = walk(path)
        lo, hi = share_records(pl[(rank, fi)], uoff, starts, len(u))
        parts.append(soa_of(u, starts, lo, hi))
        n_mine += hi - lo
    soa = tuple(np.concatenate([p[k] for p in parts]) for k in range(5))
    r = _oracle(t, rl, rows, soa)
    # wrap check: every rank's first coverage cell starts near 2^32 and its cnt[12] near 2^64
    r["cov"][0] += np.uint32(0xFFFFFFF0)
    r["cnt"][12] += np.uint64(0xFFFFFFFFFFFFFF00)
    tens = [torch.from_numpy(_signed(r[k])) for k in KEYS]
    for x in tens:
        if os.environ.get("ITX_TEST_EXCHANGE") == "reduce":
            dist.reduce(x, dst=0, op=dist.ReduceOp.SUM)         # what the command does: only the writing rank gets the sums
        else:
            dist.all_reduce(x, op=dist.ReduceOp.SUM)
    cnt = torch.tensor([n_mine], dtype=torch.int64)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    if rank == 0:
        q.put({k: r[k].copy() for k in KEYS} | {"records": int(cnt.item()), "mine": n_mine})
    dist.barrier()
    dist.destroy_process_group()


def _make_bams(tmp, n_files):
    files = []
    for k in range(n_files):
        r = synth.make_reads(300 + k, CHROMS, 9000 + 2500 * k, read_len=(30, 140), paired_frac=0.0, unmapped_frac=0.02)
        path = os.path.join(tmp, f"reads{k}.bam")
        synth.write_bam(path, r, with_seq=True, block=0x3000)        # many small blocks: the split points fall all over the file
        files.append(path)
    return files


@pytest.mark.parametrize("exchange,n_files", [("allreduce", 1), ("reduce", 2)])
def test_two_rank_shares_and_exchange_equal_single_process(exchange, n_files, monkeypatch, share_dump, tmp_path):
    monkeypatch.setenv("ITX_TEST_EXCHANGE", exchange)      # inherited by the spawned ranks
    files = _make_bams(str(tmp_path), n_files)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, q, share_dump, files)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    t, rl, rows = _table()
    parts, n_all = [], 0
    for path in files:
        u, uoff, starts = walk(path)
        parts.append(soa_of(u, starts, 0, len(starts)))
        n_all += len(starts)
    whole = _oracle(t, rl, rows, tuple(np.concatenate([p[k] for p in parts]) for k in range(5)))
    whole["cov"][0] += np.uint32((2 * 0xFFFFFFF0) & 0xFFFFFFFF)
    whole["cnt"][12] += np.uint64((2 * 0xFFFFFFFFFFFFFF00) & 0xFFFFFFFFFFFFFFFF)
    assert got["records"] == n_all and 0 < got["mine"] < n_all
    for k in KEYS:
        assert np.array_equal(got[k], whole[k]), k
    assert int(whole["cnt"][9]) > 1000


def test_shares_cover_every_record_exactly_once(share_dump, tmp_path):
    """plan_shares + find_split for 1, 2, 3 and 8 ranks over one file and over a list of three: the ranks' record ranges
    are disjoint, in order, and together they are the files."""
    files = _make_bams(str(tmp_path), 3)
    walks = [walk(p) for p in files]
    for flist in ([files[0]], files):
        for world in (1, 2, 3, 8):
            shared, pl = plan(share_dump, flist, world)
            assert shared == 1
            for fi in range(len(flist)):
                u, uoff, starts = walks[fi]
                at = 0
                for rank in range(world):
                    lo, hi = share_records(pl[(rank, fi)], uoff, starts, len(u))
                    if hi > lo:
                        assert lo == at, (world, fi, rank, lo, at)
                        at = hi
                assert at == len(starts), (world, fi)


def test_small_inputs_are_not_shared(share_dump, tmp_path):
    """less than the minimum share per rank (ITX_SPLIT_MIN in the command): rank 0 takes everything"""
    files = _make_bams(str(tmp_path), 1)
    shared, pl = plan(share_dump, files, 2, min_share=1 << 30)
    assert shared == 0
    assert pl[(0, 0)]["lo"] == 0 and pl[(0, 0)]["hi"] == SIZE_MAX and pl[(1, 0)]["lo"] == pl[(1, 0)]["hi"]
