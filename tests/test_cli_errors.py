"""CPU-side checks of the drop-in boundary's error conventions (stat.c / filter.c / iteres.c of the reference):
usage text and exit code 1 for usage errors, exit code 255 (exit(-1)) with a message for fatal ones — and, without
a GPU, a loud failure instead of any CPU computation."""
import os
import subprocess

import pytest

from iteres_amd import build, engine as eng


@pytest.fixture(scope="module")
def exe():
    lib, exe = build.build_all()
    assert exe and os.path.exists(exe)
    return exe


def run(exe, *args, cwd=None):
    return subprocess.run([exe, *args], capture_output=True, text=True, cwd=cwd, timeout=120)


def test_dispatch_and_usage(exe):
    r = run(exe)
    assert r.returncode == 1 and "Program: iteres (repeat analysis utils from Wang lab)" in r.stderr and "Version: 0.3.3-r123" in r.stderr
    r = run(exe, "bogus")
    assert r.returncode == 1 and "[iteres] unrecognized command 'bogus'" in r.stderr
    r = run(exe, "stat")
    assert r.returncode == 1 and "Usage:   iteres stat [options]" in r.stderr
    r = run(exe, "stat", "-h")
    assert r.returncode == 1 and "-E       extend reads to represent fragment [150]" in r.stderr
    r = run(exe, "filter", "a", "b")
    assert r.returncode == 1 and "Usage:   iteres filter [options]" in r.stderr


def test_fatal_errors_exit_255(exe, tmp_path):
    r = run(exe, "stat", "-N", "7", "a", "b", "c", "d", cwd=tmp_path)
    assert r.returncode == 255 and "Wrong normalization method specified" in r.stderr
    r = run(exe, "filter", "-n", "AluY", "-c", "SINE", "a", "b", "c", "d", cwd=tmp_path)
    assert r.returncode == 255 and "Please specify only one filter, either -n, -c or -f." in r.stderr
    r = run(exe, "stat", "nochrom.sizes", "b", "c", "d", cwd=tmp_path)
    assert r.returncode == 255 and "Couldn't open nochrom.sizes" in r.stderr


def test_inputs_parse_then_fail_loudly_without_gpu(exe, tmp_path):
    """Size files and rmsk load on the CPU (host logic); the record loop needs the GPU and must say so."""
    if eng.load().itx_device_count() > 0:
        pytest.skip("a GPU is present")
    (tmp_path / "c.sizes").write_text("chr1\t1000\n")
    (tmp_path / "r.sizes").write_text("AluY\t311\n")
    (tmp_path / "rmsk.txt").write_text("585\t1\t0\t0\t0\tchr1\t10\t200\t-800\t+\tAluY\tSINE\tAlu\t1\t190\t-121\t1\n")
    (tmp_path / "x.sam").write_text("@SQ\tSN:chr1\tLN:1000\nr0\t0\tchr1\t20\t30\t10M\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\n")
    r = run(exe, "stat", "-S", "c.sizes", "r.sizes", "rmsk.txt", "x.sam", cwd=tmp_path)
    assert r.returncode == 255
    assert "* Total 1 repeats found." in r.stderr and "no usable MI355X" in r.stderr
    assert not any(p.name.endswith(".stat") for p in tmp_path.iterdir())
    # a row past its chromosome end is what binKeeperAdd aborts on
    (tmp_path / "bad.txt").write_text("585\t1\t0\t0\t0\tchr1\t10\t2000\t0\t+\tAluY\tSINE\tAlu\t1\t190\t-121\t1\n")
    r = run(exe, "stat", "-S", "c.sizes", "r.sizes", "bad.txt", "x.sam", cwd=tmp_path)
    assert r.returncode == 255 and "(10 2000) out of range (0 1000) in binKeeperAdd" in r.stderr
