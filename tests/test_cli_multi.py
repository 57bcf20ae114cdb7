"""The command as a multi-GPU job (one process per GPU, iteres_amd/host/multi.c + stream.c), rehearsed on ONE GPU: the
ranks share device 0 (ITX_GPU_MAP=0,0,...) and hand their partials over through files instead of RCCL (which wants a GPU per
rank) — everything else is the code an 8-GPU node runs: the ranks the command starts itself (ITX_GPUS) or a launcher
starts (ITX_RANK / ITX_WORLD), the shares of the compressed bytes, split points guessed by one rank and verified by the rank
before it, the fall-back to one rank when a guess does not hold, the reduced partial turned into the reference's files.
Every run must reproduce the reference's files byte for byte (tests/golden) or the one-rank run's."""
import filecmp
import os
import struct
import subprocess
import time

import numpy as np
import pytest

import goldencase as gc
import refio
from iteres_amd import build, engine as eng, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def exe():
    lib, exe = build.build_all()
    assert exe and os.path.exists(exe)
    return exe


def _ranks_env(n, **kw):
    e = dict(os.environ, ITX_GPUS=str(n), ITX_GPU_MAP=",".join("0" for _ in range(n)), ITX_SPLIT_MIN="1", ITX_TIMING="1", ITX_COMM_TIMEOUT="120")
    e.update(kw)
    return e


BAM_RUNS = [(c, r) for c, r in gc.list_runs() if gc.manifest_run(c, r)["cmd"] in ("stat", "filter") and gc.manifest_run(c, r)["aln"].endswith(".bam")]


@pytest.mark.parametrize("case,run_name", BAM_RUNS)
def test_ranks_reproduce_reference_files(case, run_name, exe, tmp_path):
    """Every golden option set that reads a BAM, as a job of 3 ranks with the decoder cutting the file into small chunks:
    the reference's files, byte for byte. Option sets that are order-dependent (-R, bed files, -r) must quietly stay with
    one rank."""
    run = gc.manifest_run(case, run_name)
    src = os.path.join(gc.GOLDEN, case, "in")
    paths = [refio.materialise(src, n, str(tmp_path)) for n in ["chrom.sizes", "rep.sizes", "rmsk.txt", run["aln"]]]
    work = tmp_path / "out"
    work.mkdir()
    pr = subprocess.run([exe, run["cmd"]] + run["opts"] + ["-o", run["prefix"]] + paths, cwd=work, capture_output=True, text=True, timeout=600,
                        env=_ranks_env(3, ITX_BGZF_CHUNK="40000"))
    assert pr.returncode == run["rc"], pr.stderr[-2000:]
    for fn in run["files"]:
        want = refio.read_bytes(os.path.join(gc.GOLDEN, case, run_name, fn))
        got_path = work / fn
        assert got_path.exists(), f"{fn} missing; stderr: {pr.stderr[-1500:]}"
        assert got_path.read_bytes() == want, f"{case}/{run_name}/{fn} differs"
    order_dependent = any(o in run["opts"] for o in ("-R", "-B", "-V", "-r"))
    assert ("exchange (files)" in pr.stderr) == (not order_dependent), pr.stderr[-1500:]


@pytest.fixture(scope="module")
def big_case(tmp_path_factory):
    """a table and a 600 k-read BAM with sequence, big enough for many shares and chunks"""
    d = tmp_path_factory.mktemp("multi")
    chroms = [("chr1", 60_000_000), ("chr2", 35_000_000), ("chrM", 16_569)]
    t = synth.make_table(81, chroms, 80_000, n_names=500, n_fams=30, n_clas=10, overlap_frac=0.05, shuffle_frac=0.02)
    synth.write_sizes(str(d / "chrom.sizes"), chroms)
    synth.write_sizes(str(d / "rep.sizes"), t.rep_len.items())
    synth.write_rmsk(str(d / "rmsk.txt"), t)
    mk = os.path.join(os.path.dirname(build.HERE), "tools", "mkbam")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-o", mk, os.path.join(os.path.dirname(build.HERE), "tools", "mkbam.c"), "-lz", "-ldl"])
    for name, n, seed, xa in (("a.bam", 600_000, 5, 0), ("b.bam", 250_000, 6, 0), ("c.bam", 90_000, 7, 0), ("xa.bam", 300_000, 8, 400)):
        subprocess.check_call([mk, str(d / "chrom.sizes"), str(n), str(d / name), "60", str(seed), str(xa)])
    return d


def _run(exe, head, d, aln, out, env):
    os.makedirs(out, exist_ok=True)
    pr = subprocess.run([exe] + head + ["-o", "out", str(d / "chrom.sizes"), str(d / "rep.sizes"), str(d / "rmsk.txt"), aln], cwd=out, capture_output=True,
                        text=True, timeout=600, env=env)
    assert pr.returncode == 0, pr.stderr[-2500:]
    return pr


def _same_dir(a, b):
    names = sorted(os.listdir(a))
    assert names and names == sorted(os.listdir(b))
    for fn in names:
        assert filecmp.cmp(os.path.join(a, fn), os.path.join(b, fn), shallow=False), fn


@pytest.mark.parametrize("head", [["stat", "-w"], ["filter", "-n", "Rep3"], ["stat", "-w", "-E", "0", "-Q", "30"]])
def test_shares_of_one_file_and_of_a_file_list(head, exe, big_case):
    """2, 3 and 4 ranks over ONE file and over a LIST of three files of different sizes (shares then start and end inside
    different files, some ranks get pieces of two files): same files as the one-rank run, whatever the chunking."""
    d = big_case
    tag = "_".join(head).replace("-", "")
    lst = ",".join(str(d / n) for n in ("a.bam", "b.bam", "c.bam"))
    for aln, what in ((str(d / "a.bam"), "one"), (lst, "list")):
        if head[0] == "filter" and what == "list":
            continue                                                     # filter takes one file
        ref_dir = str(d / f"ref_{tag}_{what}")
        _run(exe, head, d, aln, ref_dir, dict(os.environ))
        for n, chunk in ((2, None), (3, "300000"), (4, "90000")):
            out = str(d / f"r{n}_{tag}_{what}")
            pr = _run(exe, head, d, aln, out, _ranks_env(n, **({"ITX_BGZF_CHUNK": chunk} if chunk else {})))
            assert "exchange (files)" in pr.stderr and "share boundary" not in pr.stderr, pr.stderr[-1500:]
            _same_dir(ref_dir, out)


def test_xa_veto_shards_by_record(exe, big_case):
    """The XA veto looks at one record and the table, nothing else: it shards like the rest (its counter travels with the
    partial). A BAM with XA tags on 40 % of the reads, 3 ranks vs one."""
    d = big_case
    ref_dir, out = str(d / "xa_ref"), str(d / "xa_r3")
    _run(exe, ["stat", "-w"], d, str(d / "xa.bam"), ref_dir, dict(os.environ))
    pr = _run(exe, ["stat", "-w"], d, str(d / "xa.bam"), out, _ranks_env(3, ITX_BGZF_CHUNK="200000"))
    assert "exchange (files)" in pr.stderr
    _same_dir(ref_dir, out)
    rep = open(os.path.join(out, "out.iteres.report")).read()
    assert "different subfamilies: 0" not in rep                           # the veto did fire


def test_ranks_started_by_a_launcher(exe, big_case):
    """What bench.py does under torch.distributed.run: the ranks are separate processes somebody else started, told who
    they are through ITX_RANK / ITX_WORLD / ITX_DEVICE / ITX_COMM_ID / ITX_EXCHANGE; rank 0 writes the files."""
    d = big_case
    ref_dir = str(d / "launch_ref")
    _run(exe, ["stat", "-w"], d, str(d / "a.bam"), ref_dir, dict(os.environ))
    world = 3
    procs = []
    for r in range(world):
        out = str(d / f"launch_rank{r}")
        os.makedirs(out, exist_ok=True)
        env = dict(os.environ, ITX_RANK=str(r), ITX_WORLD=str(world), ITX_DEVICE="0", ITX_COMM_ID=str(d / "launch.id"), ITX_EXCHANGE="file", ITX_SPLIT_MIN="1",
                   ITX_COMM_TIMEOUT="120")
        procs.append(subprocess.Popen([exe, "stat", "-w", "-o", "out", str(d / "chrom.sizes"), str(d / "rep.sizes"), str(d / "rmsk.txt"), str(d / "a.bam")], cwd=out,
                                      env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
    errs = [p.communicate(timeout=600)[1] for p in procs]
    assert all(p.returncode == 0 for p in procs), errs
    _same_dir(ref_dir, str(d / "launch_rank0"))
    for r in range(1, world):
        assert os.listdir(str(d / f"launch_rank{r}")) == []                 # only rank 0 writes
        assert errs[r] == ""                                                  # and only rank 0 talks


def test_false_split_point_falls_back_to_one_rank(exe, tmp_path):
    """A split point is a guess (8 well-formed records in a row). Here every record carries, inside a byte-array tag, forty
    well-formed DECOY records — nine tenths of the file are decoys, and BGZF blocks start inside them: the guesses land on
    decoys (tests/test_host_reader.py shows it for this very file), the rank before finds that its record chain does not
    arrive there, and the job is done again by one rank — same files."""
    chroms = [("c1", 3_000_000)]
    t = synth.make_table(91, chroms, 4_000, n_names=60, n_fams=10, n_clas=4)
    synth.write_sizes(str(tmp_path / "chrom.sizes"), chroms)
    synth.write_sizes(str(tmp_path / "rep.sizes"), t.rep_len.items())
    synth.write_rmsk(str(tmp_path / "rmsk.txt"), t)
    r = synth.make_reads(92, chroms, 12_000, read_len=(30, 60))
    fake = struct.pack("<iiiIIiiii", 40, 0, 5, 2 | (30 << 8), 0, 0, -1, -1, 0) + b"a\0" + bytes(6)
    decoy = (fake * 40 + struct.pack("<i", 33) + bytes(range(40, 80))).hex()
    r.aux = [[f"ZZ:B:{decoy}"] for _ in range(len(r))]
    bam = str(tmp_path / "decoy.bam")
    synth.write_bam(bam, r, with_seq=True)

    class D:                                                               # _run takes a directory-like with the three tables
        def __truediv__(self, n):
            return tmp_path / n
    ref_dir, out = str(tmp_path / "ref"), str(tmp_path / "r4")
    _run(exe, ["stat", "-w"], D(), bam, ref_dir, dict(os.environ))
    pr = _run(exe, ["stat", "-w"], D(), bam, out, _ranks_env(4))
    assert "share boundary was not a record start" in pr.stderr, pr.stderr[-1500:]
    _same_dir(ref_dir, out)


def test_rccl_bring_up_with_one_rank(tmp_path):
    """RCCL refuses two ranks on one GPU, so on a one-GPU box the ncclReduce path can only be walked with ONE rank
    (ITX_COMM_SELFTEST): library loaded, entry points found, id handed over through the file, communicator made, both
    reduces and the host vector through it — the buffers must come back unchanged."""
    import ctypes as C
    import torch
    L = eng.load()
    os.environ["ITX_COMM_SELFTEST"] = "1"
    try:
        h = C.c_void_p()
        rc = L.itx_comm_create(0, 1, 0, str(tmp_path / "self.id").encode(), 0, C.byref(h))
        assert rc == 0, L.itx_last_error()
        a = torch.arange(1, 100_001, dtype=torch.int64, device="cuda:0")
        b = torch.arange(7, 300_007, dtype=torch.int32, device="cuda:0")
        meta = np.array([5, 6, 7, 8], np.uint64)
        torch.cuda.synchronize()
        rc = L.itx_comm_reduce_sum(h, a.data_ptr(), a.numel(), b.data_ptr(), b.numel(), meta.ctypes.data_as(C.c_void_p), 4, None)
        assert rc == 0, L.itx_last_error()
        assert int(a.sum()) == 100_000 * 100_001 // 2 and int(b[0]) == 7 and int(b[-1]) == 300_006
        assert list(meta) == [5, 6, 7, 8]
        L.itx_comm_destroy(h)
    finally:
        os.environ.pop("ITX_COMM_SELFTEST", None)


@pytest.mark.parametrize("head", [["stat", "-w"], ["filter", "-n", "Rep3"]])
def test_command_walks_the_rccl_exchange_with_one_rank(head, exe, big_case, tmp_path):
    """ITX_COMM_SELFTEST + ITX_WORLD=1: the command as a job of ONE rank that still makes its RCCL communicator beside the
    scan (the early thread of host/stream.c), exports its partial, reduces it through ncclReduce and finishes from the reduced
    buffers — the N > 1 path end to end as far as a one-GPU box can take it. Same files as the plain run."""
    d = big_case
    aln = str(d / "a.bam")
    ref_dir, out = str(tmp_path / "plain"), str(tmp_path / "self")
    _run(exe, head, d, aln, ref_dir, dict(os.environ))
    env = dict(os.environ, ITX_RANK="0", ITX_WORLD="1", ITX_DEVICE="0", ITX_COMM_ID=str(tmp_path / "self.id"), ITX_EXCHANGE="rccl",
               ITX_COMM_SELFTEST="1", ITX_TIMING="1", ITX_COMM_TIMEOUT="120")
    pr = _run(exe, head, d, aln, out, env)
    assert "exchange (RCCL)" in pr.stderr, pr.stderr[-1500:]
    _same_dir(ref_dir, out)
    # and with the communicator made at the exchange instead (the way it was before): same again
    out2 = str(tmp_path / "self_late")
    pr = _run(exe, head, d, aln, out2, dict(env, ITX_NO_EARLY_COMM="1", ITX_COMM_ID=str(tmp_path / "self2.id")))
    assert "exchange (RCCL)" in pr.stderr, pr.stderr[-1500:]
    _same_dir(ref_dir, out2)


def test_rank0_without_rccl_sends_the_others_to_the_files(exe, big_case, tmp_path):
    """Two ranks started by hand (the way bench.py starts them), RCCL asked for. Rank 0 cannot load the library
    (ITX_COMM_NO_RCCL): instead of the communicator id the waiting rank finds a note, gives RCCL up at once (not after its
    timeout) and both hand their partials over through files. Same files as one rank."""
    d = big_case
    aln = str(d / "a.bam")
    ref_dir = str(tmp_path / "plain")
    _run(exe, ["stat", "-w"], d, aln, ref_dir, dict(os.environ))
    args = [exe, "stat", "-w", "-o", "out", str(d / "chrom.sizes"), str(d / "rep.sizes"), str(d / "rmsk.txt"), aln]
    procs = []
    t0 = time.time()
    for r in range(2):
        out = tmp_path / f"rank{r}"
        out.mkdir()
        env = dict(os.environ, ITX_RANK=str(r), ITX_WORLD="2", ITX_DEVICE="0", ITX_COMM_ID=str(tmp_path / "job.id"), ITX_EXCHANGE="rccl",
                   ITX_SPLIT_MIN="1", ITX_TIMING="1", ITX_COMM_TIMEOUT="100")
        if r == 0:
            env["ITX_COMM_NO_RCCL"] = "1"
        procs.append(subprocess.Popen(args, cwd=out, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
    errs = [p.communicate(timeout=300)[1] for p in procs]
    assert [p.returncode for p in procs] == [0, 0], errs
    assert time.time() - t0 < 60                                    # nobody sat out a timeout
    assert "exchange (files)" in errs[0] and "could not make a communicator id" in errs[1], errs
    _same_dir(ref_dir, str(tmp_path / "rank0"))


def test_a_later_rank_without_rccl_takes_every_rank_to_the_files(exe, big_case, tmp_path):
    """The other way round: rank 0 has its library and is already inside ncclCommInitRank — which has no timeout — when
    rank 1 finds it cannot load RCCL. Every rank leaves an ok / fail marker next to the communicator id when its attempt has
    ended and RCCL is used only if all say ok: rank 0 sees rank 1's "fail", leaves its communicator thread behind and both
    exchange through files, at once. Same files as one rank; the markers are gone afterwards."""
    d = big_case
    aln = str(d / "a.bam")
    ref_dir = str(tmp_path / "plain")
    _run(exe, ["stat", "-w"], d, aln, ref_dir, dict(os.environ))
    args = [exe, "stat", "-w", "-o", "out", str(d / "chrom.sizes"), str(d / "rep.sizes"), str(d / "rmsk.txt"), aln]
    procs = []
    t0 = time.time()
    for r in range(2):
        out = tmp_path / f"rank{r}"
        out.mkdir()
        env = dict(os.environ, ITX_RANK=str(r), ITX_WORLD="2", ITX_DEVICE="0", ITX_COMM_ID=str(tmp_path / "job.id"), ITX_EXCHANGE="rccl",
                   ITX_SPLIT_MIN="1", ITX_TIMING="1", ITX_COMM_TIMEOUT="100")
        if r == 1:
            env["ITX_COMM_NO_RCCL"] = "1"
        procs.append(subprocess.Popen(args, cwd=out, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
    errs = [p.communicate(timeout=300)[1] for p in procs]
    assert [p.returncode for p in procs] == [0, 0], errs
    assert time.time() - t0 < 60                                    # nobody sat out a timeout
    assert "exchange (files)" in errs[0] and "cannot load librccl" in errs[1], errs
    _same_dir(ref_dir, str(tmp_path / "rank0"))
    assert not [f for f in os.listdir(tmp_path) if f.startswith("job.id")], os.listdir(tmp_path)


# ---- for the first box with more than one GPU: two RCCL ranks really meet (skipped where one device is visible) ---------------
def _n_devices():
    import torch
    return torch.cuda.device_count()                                       # (counting devices does not initialise the GPU)


need2 = pytest.mark.skipif(_n_devices() < 2, reason="needs at least two visible GPUs (two RCCL ranks cannot share a device)")


@need2
@pytest.mark.parametrize("head", [["stat", "-w"], ["filter", "-n", "Rep3"]])
@pytest.mark.parametrize("gpus", ["2", "all"])
def test_rccl_exchange_between_real_ranks(head, gpus, exe, big_case, tmp_path):
    """ITX_GPUS=2 / all on a multi-GPU box: the command starts its ranks itself, one per device, every rank makes its RCCL
    communicator beside the scan, the partials meet in ncclReduce on rank 0 — asserted from the timing line — and the files are
    the one-rank run's, byte for byte (stat over a list of files as well: shares that start and end inside different files)."""
    d = big_case
    lst = ",".join(str(d / n) for n in ("a.bam", "b.bam", "c.bam"))
    for aln, what in ((str(d / "a.bam"), "one"), (lst, "list")):
        if head[0] == "filter" and what == "list":
            continue
        ref_dir, out = str(tmp_path / f"plain_{what}"), str(tmp_path / f"rccl_{what}")
        _run(exe, head, d, aln, ref_dir, dict(os.environ, ITX_GPUS="1"))
        env = dict(os.environ, ITX_GPUS=gpus, ITX_SPLIT_MIN="1", ITX_TIMING="1", ITX_COMM_TIMEOUT="120")
        env.pop("ITX_GPU_MAP", None)
        pr = _run(exe, head, d, aln, out, env)
        assert "exchange (RCCL)" in pr.stderr and "share boundary" not in pr.stderr, pr.stderr[-2000:]
        _same_dir(ref_dir, out)


@need2
def test_rccl_exchange_reproduces_reference_files(exe, tmp_path):
    """... and against the reference's own files: BASELINE configs[0] (the 100 k-read chr22 case) as a job of two RCCL ranks."""
    case, run_name = next((c, r) for c, r in BAM_RUNS if c.startswith("cfg1") and gc.manifest_run(c, r)["cmd"] == "stat"
                          and not any(o in gc.manifest_run(c, r)["opts"] for o in ("-R", "-B", "-V")))
    run = gc.manifest_run(case, run_name)
    src = os.path.join(gc.GOLDEN, case, "in")
    paths = [refio.materialise(src, n, str(tmp_path)) for n in ["chrom.sizes", "rep.sizes", "rmsk.txt", run["aln"]]]
    work = tmp_path / "out"
    work.mkdir()
    env = dict(os.environ, ITX_GPUS="2", ITX_SPLIT_MIN="1", ITX_TIMING="1", ITX_COMM_TIMEOUT="120", ITX_BGZF_CHUNK="400000")
    env.pop("ITX_GPU_MAP", None)
    pr = subprocess.run([exe, run["cmd"]] + run["opts"] + ["-o", run["prefix"]] + paths, cwd=work, capture_output=True, text=True, timeout=600, env=env)
    assert pr.returncode == run["rc"], pr.stderr[-2000:]
    assert "exchange (RCCL)" in pr.stderr, pr.stderr[-2000:]
    for fn in run["files"]:
        assert (work / fn).read_bytes() == refio.read_bytes(os.path.join(gc.GOLDEN, case, run_name, fn)), fn


@need2
def test_two_launcher_ranks_meet_in_rccl(exe, big_case, tmp_path):
    """The way bench.py --gpus N runs the command: ranks started by somebody else (ITX_RANK / ITX_WORLD / ITX_DEVICE), each on its
    own device, RCCL asked for: rank 0 writes the one-rank run's files, rank 1 is silent."""
    d = big_case
    aln = str(d / "a.bam")
    ref_dir = str(tmp_path / "plain")
    _run(exe, ["stat", "-w"], d, aln, ref_dir, dict(os.environ, ITX_GPUS="1"))
    args = [exe, "stat", "-w", "-o", "out", str(d / "chrom.sizes"), str(d / "rep.sizes"), str(d / "rmsk.txt"), aln]
    procs = []
    for r in range(2):
        out = tmp_path / f"rank{r}"
        out.mkdir()
        env = dict(os.environ, ITX_RANK=str(r), ITX_WORLD="2", ITX_DEVICE=str(r), ITX_COMM_ID=str(tmp_path / "job.id"), ITX_EXCHANGE="rccl",
                   ITX_SPLIT_MIN="1", ITX_TIMING="1", ITX_COMM_TIMEOUT="120")
        procs.append(subprocess.Popen(args, cwd=out, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
    errs = [p.communicate(timeout=300)[1] for p in procs]
    assert [p.returncode for p in procs] == [0, 0], errs
    assert "exchange (RCCL)" in errs[0], errs[0][-1500:]
    _same_dir(ref_dir, str(tmp_path / "rank0"))
    assert os.listdir(str(tmp_path / "rank1")) == []


# ---- records parsed ahead of the table by the helper thread (host/stream.c prefetch_records) ---------------------------------
@pytest.mark.parametrize("head", [["stat", "-w"], ["filter", "-n", "Rep3"], ["stat", "-w", "-x", "-E", "0"]])
def test_records_parsed_ahead_of_the_table(head, exe, big_case, tmp_path):
    """While the rmsk file is parsed and the table built the helper thread parses the decoded windows and keeps their records in a
    backlog in HBM; run_stream submits the backlog first. Forced here on a small input (ITX_PREFETCH_MIN=0, many small windows,
    the table "taking" 1.5 s): some windows do go through the backlog, and the files are those of a run without it — single
    file, file list (only the first file is parsed ahead), and a BAM with XA tags (its windows are left to the loop when the
    veto is on, taken with -x)."""
    d = big_case
    lst = ",".join(str(d / n) for n in ("a.bam", "b.bam"))
    for aln, what in ((str(d / "a.bam"), "one"), (lst, "list"), (str(d / "xa.bam"), "xa")):
        if head[0] == "filter" and what == "list":
            continue
        plain, ahead = str(tmp_path / f"plain_{what}"), str(tmp_path / f"ahead_{what}")
        _run(exe, head, d, aln, plain, dict(os.environ, ITX_GPUS="1", ITX_NO_PREFETCH="1"))
        pr = _run(exe, head, d, aln, ahead, dict(os.environ, ITX_GPUS="1", ITX_TIMING="1", ITX_PREFETCH_MIN="0", ITX_PREFETCH_HOLD_MS="1500", ITX_BGZF_CHUNK="400000"))
        _same_dir(plain, ahead)
        import re
        m = re.search(r"parsed ahead of the table by the helper thread: (\d+) records of (\d+) windows", pr.stderr)
        veto_keeps_them = what == "xa" and head[0] == "stat" and "-x" not in head
        if veto_keeps_them:
            assert m is None or int(m.group(2)) == 0, pr.stderr[-1500:]
        else:
            assert m and int(m.group(2)) >= 2 and int(m.group(1)) > 50_000, pr.stderr[-1500:]
