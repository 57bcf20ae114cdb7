"""Where the host program runs (iteres_amd/host/numa.c): ITX_CPUS is obeyed, without a GPU's memory node to go by nothing
changes, and on a GPU box the process ends up on the processors of the node its GPU hangs off — with the same files."""
import os
import re
import subprocess

import pytest

import goldencase as gc
import refio
from iteres_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "iteres_amd", "host")


def _report(exe, env, args=("stat",)):
    pr = subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=120, env=dict(os.environ, ITX_NUMA_REPORT="1", **env))
    m = re.search(r"\[itx numa\] (.*): (\d+) processors, (-?\d+) \.\. (-?\d+)", pr.stderr)
    assert m, pr.stderr[-500:]
    return m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))


@pytest.fixture(scope="module")
def exe_cpu():
    # the host program needs the engine's library to link: build both when a compiler for it is at hand, else use what is there
    exe = os.path.join(HOST, "iteres")
    if not os.path.exists(exe):
        build.build_all()
    assert os.path.exists(exe)
    return exe


def test_itx_cpus_is_obeyed(exe_cpu):
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip("one processor")
    a, b = allowed[0], allowed[1]
    how, n, lo, hi = _report(exe_cpu, {"ITX_CPUS": f"{a},{b}"})
    assert (how, n, lo, hi) == ("ITX_CPUS", 2, a, b)
    how, n, lo, hi = _report(exe_cpu, {"ITX_CPUS": f"{a}-{a}"})
    assert (how, n, lo, hi) == ("ITX_CPUS", 1, a, a)
    # nonsense leaves the process where it was
    how, n, _, _ = _report(exe_cpu, {"ITX_CPUS": "x"})
    assert n == len(allowed)


def test_switched_off_or_no_gpu_leaves_the_affinity_alone(exe_cpu):
    allowed = sorted(os.sched_getaffinity(0))
    envs = [{"ITX_NUMA": "0"}]
    if not os.path.isdir("/sys/class/kfd/kfd/topology/nodes") or _gpu_node_cpus() is None:
        envs.append({})                                        # no GPU's node to go by (the CPU suite's container)
    for env in envs:
        how, n, lo, hi = _report(exe_cpu, env)
        assert (how, n, lo, hi) == ("as started", len(allowed), allowed[0], allowed[-1])


def _gpu_node_cpus():
    """the node of the first GPU this process may open, by the same files numa.c reads"""
    base = "/sys/class/kfd/kfd/topology/nodes"
    for k in sorted(int(x) for x in os.listdir(base)):
        try:
            props = dict(ln.split()[:2] for ln in open(f"{base}/{k}/properties") if len(ln.split()) >= 2)
        except OSError:                 # another tenant's GPU: the driver does not let us read it
            continue
        if int(props.get("simd_count", 0)) <= 0 or "drm_render_minor" not in props:
            continue
        minor = int(props["drm_render_minor"])
        try:
            os.close(os.open(f"/dev/dri/renderD{minor}", os.O_RDWR))
        except OSError:
            continue
        node = int(open(f"/sys/class/drm/renderD{minor}/device/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        return cpus
    return None


@pytest.mark.gpu
def test_the_process_stays_on_its_gpus_node_and_writes_the_same_files(tmp_path):
    lib, exe = build.build_all()
    want = _gpu_node_cpus()
    allowed = set(os.sched_getaffinity(0))
    if not want or not (want & allowed) or (want & allowed) == allowed:
        pytest.skip("one memory node, or the GPU's node is not known here")
    how, n, lo, hi = _report(exe, {})
    assert how == "the GPU's memory node" and n == len(want & allowed) and lo == min(want & allowed) and hi == max(want & allowed)
    # and a whole command from there: the reference's files
    run = gc.manifest_run("mid", "stat_default")
    src = os.path.join(gc.GOLDEN, "mid", "in")
    paths = [refio.materialise(src, nm, str(tmp_path)) for nm in ["chrom.sizes", "rep.sizes", "rmsk.txt", run["aln"]]]
    for name, env in (("bound", {}), ("free", {"ITX_NUMA": "0"})):
        work = tmp_path / name
        work.mkdir()
        pr = subprocess.run([exe, run["cmd"]] + run["opts"] + ["-o", run["prefix"]] + paths, cwd=work, capture_output=True, text=True, timeout=600,
                            env=dict(os.environ, ITX_NUMA_REPORT="1", **env))
        assert pr.returncode == run["rc"], pr.stderr[-2000:]
        assert ("the GPU's memory node" in pr.stderr) == (name == "bound")
        for fn in run["files"]:
            assert (work / fn).read_bytes() == refio.read_bytes(os.path.join(gc.GOLDEN, "mid", "stat_default", fn)), f"{name}: {fn}"
