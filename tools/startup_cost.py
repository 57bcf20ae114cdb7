#!/usr/bin/env python3
"""Scratch: where the wall time of the command goes outside main(): start-up (exec, dynamic loading) and tear-down (the
kernel releasing device and page-locked memory after _exit). Needs bench.py's inputs (tools/ab_cli.py makes them)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
exe = bench.OURS
t0 = time.perf_counter(); subprocess.run([exe], capture_output=True); print(f"no arguments (load + usage): {time.perf_counter() - t0:.3f} s")
t0 = time.perf_counter(); subprocess.run([exe], capture_output=True); print(f"again: {time.perf_counter() - t0:.3f} s")
wd = sys.argv[1]
for env_extra in ({}, {"ITX_RESERVE_WINDOWS": "8"}, {"ITX_NO_RESERVE": "1"}):
    env = dict(os.environ, OMP_NUM_THREADS="16", ITX_TIMING="1", ITX_GPUS="1", **env_extra)
    os.makedirs(os.path.join(wd, "sc"), exist_ok=True)
    t0 = time.time()
    p = subprocess.Popen([exe] + bench.base_args(wd) + [os.path.join(wd, "reads.bam")], cwd=os.path.join(wd, "sc"), env=env, stderr=subprocess.PIPE, text=True)
    err = p.stderr.read()          # EOF when the process has closed stderr (at _exit)
    t_eof = time.time()
    p.wait()
    t_reaped = time.time()
    inside = [ln for ln in err.split("\n") if "main() entered" in ln]
    print(env_extra, f"wall {t_reaped - t0:.3f} s; stderr closed at {t_eof - t0:.3f} s; reaped {t_reaped - t_eof:.3f} s later;", inside)
# the same with a GPU context held by THIS process (what bench.py did while it ran the command)
import torch
torch.cuda.set_device(0)
torch.cuda.synchronize()
x = torch.zeros(1, device="cuda:0")
for env_extra in ({}, {}):
    env = dict(os.environ, OMP_NUM_THREADS="16", ITX_TIMING="1", ITX_GPUS="1", **env_extra)
    t0 = time.time()
    p = subprocess.Popen([exe] + bench.base_args(wd) + [os.path.join(wd, "reads.bam")], cwd=os.path.join(wd, "sc"), env=env, stderr=subprocess.PIPE, text=True)
    err = p.stderr.read()
    t_eof = time.time()
    p.wait()
    t_reaped = time.time()
    inside = [ln for ln in err.split("\n") if "main() entered" in ln or "HIP runtime" in ln]
    print("parent holds a context:", f"wall {t_reaped - t0:.3f} s; stderr closed at {t_eof - t0:.3f} s; reaped {t_reaped - t_eof:.3f} s later;", inside)
