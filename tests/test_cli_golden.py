"""End-to-end drop-in test (GPU): the C host program `iteres` (iteres_amd/host), driving the HIP engine through the
C ABI, must reproduce the reference's output FILES byte for byte on every golden case — same options, same file
names (tests/golden/*/manifest.json holds what oracle/_ref/iteres wrote)."""
import os
import subprocess

import pytest

import goldencase as gc
import refio
from iteres_amd import build

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def exe():
    lib, exe = build.build_all()
    assert exe and os.path.exists(exe)
    return exe


@pytest.mark.parametrize("case,run_name", gc.list_runs())
def test_cli_matches_reference_files(case, run_name, exe, tmp_path):
    run = gc.manifest_run(case, run_name)
    src = os.path.join(gc.GOLDEN, case, "in")
    names = ["chrom.sizes", "rep.sizes", "rmsk.txt", run["aln"]]
    paths = [refio.materialise(src, n, str(tmp_path)) for n in names]
    work = tmp_path / "out"
    work.mkdir()
    pr = subprocess.run([exe, run["cmd"]] + run["opts"] + ["-o", run["prefix"]] + paths, cwd=work, capture_output=True, text=True,
                        timeout=600)
    assert pr.returncode == run["rc"], pr.stderr[-2000:]
    for fn in run["files"]:
        want = refio.read_bytes(os.path.join(gc.GOLDEN, case, run_name, fn))
        got_path = work / fn
        assert got_path.exists(), f"{fn} missing; stderr: {pr.stderr[-1500:]}"
        got = got_path.read_bytes()
        assert got == want, f"{case}/{run_name}/{fn} differs"
    # bigWig (stat.c:156-158): same decoded content as the reference's file — chromosomes, sections, zoom records,
    # summary (tests/refio.py bigwig_digest; the deflate bytes depend on the zlib at hand)
    for fn, want in run.get("bigwig_sha256", {}).items():
        got_path = work / fn
        assert got_path.exists(), f"{fn} missing"
        assert refio.bigwig_digest(got_path.read_bytes()) == want, f"{case}/{run_name}/{fn}: decoded bigWig differs"
    # the banners the reference prints around the phases are part of the boundary too
    err = pr.stderr.replace("\r", "\n")
    if run["cmd"] == "cpgstat":
        for banner in ("* Start to parse the rmsk file", "* Start to parse the bedGraph file", "* Processed CpG sites:", "* CpG sites in Repeats:",
                       "* Writing stats and Wig file", "* Generating bigWig files", "* Done, time used"):
            assert banner in err
    elif run["cmd"] == "cpgfilter":
        for banner in ("* Start to parse the rmsk file", "* Start to parse the bedGraph file", "* Preparing the output file", "* Done, time used"):
            assert banner in err
        assert run["stderr_tail"][-2] in err                      # "* Total N [name] TEs have CpG score larger than T."
    elif run["cmd"] == "stat":
        for banner in ("* Provided 1 BAM/SAM file(s)", "* Parsing the rmsk file", "* Parsing the SAM/BAM file", "* Writing stats and Wig file",
                       "* Preparing report file", "* Done, time used"):
            assert banner in err
    else:
        for banner in ("* Start to parse the rmsk file", "* Start to parse the SAM/BAM file", "* Preparing the output file", "* Done, time used"):
            assert banner in err


@pytest.mark.parametrize("env", [{"ITX_BGZF_CHUNK": "30000", "ITX_HOP_PIECE": "500"}, {"ITX_BGZF_CHUNK": "3000"}, {"ITX_DEV_WINDOW_BLOCKS": "2"}, {"ITX_HOST_INFLATE": "1"},
                                 {"ITX_HOST_INFLATE": "1", "ITX_BGZF_CHUNK": "30000"}])
def test_cli_decode_paths_agree(env, exe, tmp_path):
    """The decode side of the drop-in has several routes — BGZF blocks inflated on the device (default) or by the host's
    threads, the file taken in one chunk or in hundreds (carry-over of partial blocks and records, buffer swaps), records
    located in pieces — and every one of them must produce the reference's files."""
    for case, run_name in (("mid", "stat_default"), ("sidechan", "stat_veto"), ("sidechan", "filter_R")):
        run = gc.manifest_run(case, run_name)
        if not run["aln"].endswith(".bam"):
            continue
        src = os.path.join(gc.GOLDEN, case, "in")
        d = tmp_path / f"{case}_{run_name}"
        d.mkdir()
        paths = [refio.materialise(src, n, str(d)) for n in ["chrom.sizes", "rep.sizes", "rmsk.txt", run["aln"]]]
        work = d / "out"
        work.mkdir()
        pr = subprocess.run([exe, run["cmd"]] + run["opts"] + ["-o", run["prefix"]] + paths, cwd=work, capture_output=True, text=True, timeout=600,
                            env=dict(os.environ, **env))
        assert pr.returncode == run["rc"], pr.stderr[-2000:]
        for fn in run["files"]:
            want = refio.read_bytes(os.path.join(gc.GOLDEN, case, run_name, fn))
            assert (work / fn).read_bytes() == want, f"{env}: {case}/{run_name}/{fn} differs"


def test_bigwig_files_written_side_by_side_equal_one_after_the_other(exe, tmp_path):
    """`stat` writes its two bigWig files one after the other with all of its threads; ITX_BW_PAIR=1 writes them at the same
    time, each with half of the threads (cmd_stat.c): the same bytes either way, and with one thread."""
    run = gc.manifest_run("mid", "stat_default")
    src = os.path.join(gc.GOLDEN, "mid", "in")
    paths = [refio.materialise(src, n, str(tmp_path)) for n in ["chrom.sizes", "rep.sizes", "rmsk.txt", run["aln"]]]
    seen = {}
    for name, env in (("side_by_side", {"ITX_BW_PAIR": "1"}), ("serial", {}), ("one_thread", {"OMP_NUM_THREADS": "1"})):
        work = tmp_path / name
        work.mkdir()
        pr = subprocess.run([exe, run["cmd"]] + run["opts"] + ["-o", run["prefix"]] + paths, cwd=work, capture_output=True, text=True, timeout=600,
                            env=dict(os.environ, **env))
        assert pr.returncode == run["rc"], pr.stderr[-2000:]
        seen[name] = {fn: (work / fn).read_bytes() for fn in run.get("bigwig_sha256", {})}
        assert seen[name], "the run is expected to write bigWig files"
        for fn, want in run["bigwig_sha256"].items():
            assert refio.bigwig_digest(seen[name][fn]) == want, f"{name}: {fn} differs from the reference's"
    assert seen["side_by_side"] == seen["serial"] == seen["one_thread"]
