"""Pins oracle/ against the LIVE reference on inputs nobody chose (CPU only, where oracle/_ref/iteres exists — the build
container): seeded random tables, reads and option sets go through the reference binary (the golden generator's own
run_ref, into a scratch directory laid out like tests/golden/) and through the checks of tests/test_oracle_golden.py —
report counters, the three stat files row by row, per-base coverage, per-locus counts and read lists. The committed golden
cases pin chosen corners; this sweeps combinations. Skipped where the reference binary is absent (the GPU box)."""
import importlib.util
import os
import sys

import numpy as np
import pytest

import goldencase as gc
import test_cli_vs_reference_fuzz as fz
import test_oracle_golden as og

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "iteres")


@pytest.fixture(scope="module")
def mg():
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/iteres not built (make -C oracle ref)")
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("seed", list(range(8100, 8132)))
def test_oracle_matches_reference_binary_on_a_random_case(seed, mg, tmp_path, monkeypatch):
    rng = np.random.default_rng(seed)
    _, chroms, t, r = fz._random_case(seed)
    runs = []
    for k, (cmd, opts) in enumerate(fz._random_opts(rng, t)):
        opts = [o for o in opts]
        runs.append((f"{cmd}_{k}", cmd, opts, "reads.bam"))
    monkeypatch.setattr(mg, "HERE", str(tmp_path))
    monkeypatch.setattr(gc, "GOLDEN", str(tmp_path))
    mg.emit_case("fuzz", t, r, runs, sam=False)
    for rn, cmd, opts, _ in runs:
        run = gc.manifest_run("fuzz", rn)
        if run["rc"] != 0:
            continue                          # an option set the reference refuses: nothing to compare
        (og.test_stat_counts_and_coverage if cmd == "stat" else og.test_filter_loci)("fuzz", rn)
