"""The N > 1 path on CPU (gloo, world size 2): records are sharded over ranks, every rank produces its own partial,
ONE sum-all-reduce merges them — the result must equal the single-process result. The per-rank work is done by the
oracle here (no GPU in this container); what is under test is the sharding arithmetic and the exchange, including
the mod-2^32 / mod-2^64 behaviour of unsigned sums carried in signed tensors."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from iteres_amd import dist as idist, synth  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _workload():
    from iteres_amd import engine as eng
    chroms = [("c1", 6_000_000), ("c2", 2_000_000)]
    t = synth.make_table(5, chroms, 9000, n_names=120, n_fams=15, n_clas=6, overlap_frac=0.05, inconsistent_frac=0.02)
    rl = np.array([t.rep_len.get(n, 0) for n in t.names], np.uint32)
    rows = eng.make_rows(t.chrom, t.start, t.end, t.cons_start, t.cons_end, t.rep_name, t.fam_of_row, t.cla_of_row)
    reads = synth.make_reads_soa(6, chroms, 30_001)
    return chroms, t, rl, rows, reads


def _oracle_partial(chroms, t, rl, rows, reads, lo, hi):
    from oracle import binding as orc
    tid, pos, tmpend, mapq, f5 = (a[lo:hi] for a in reads)
    ot = orc.OracleTable([s for _, s in chroms], rl, len(t.fams), len(t.clas))
    ot.add_rows(rows["chrom"], rows["start"], rows["end"], rows["cons_start"], rows["cons_end"], rows["rep"], rows["fam"], rows["cla"])
    r = ot.run({}, [0, 1], tid, pos, tmpend, mapq, np.where(f5 & 8, 16, 0).astype(np.uint16), want_hits=False)
    ot.close()
    return r


def _rank(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    chroms, t, rl, rows, reads = _workload()
    lo, hi = idist.shard_bounds(len(reads[0]), rank, world, align=1024)
    r = _oracle_partial(chroms, t, rl, rows, reads, lo, hi)
    # wrap check: start every rank's first coverage cell near 2^32 and its cnt[12] near 2^64
    r["cov"][0] += np.uint32(0xFFFFFFF0)
    r["cnt"][12] += np.uint64(0xFFFFFFFFFFFFFF00)
    keys = ("cnt", "rep_cnt", "fam_cnt", "cla_cnt", "cov", "cov_uniq")
    tens = [torch.from_numpy(idist.as_signed_view(r[k])) for k in keys]
    if os.environ.get("ITX_TEST_EXCHANGE") == "reduce":
        idist.reduce_sum_(tens, dist, dst=0)            # what bench.py does: only the writing rank gets the sums
    else:
        idist.allreduce_sum_(tens, dist)
    if rank == 0:
        q.put({k: r[k].copy() for k in keys} | {"bounds": (lo, hi)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["allreduce", "reduce"])
def test_two_rank_exchange_equals_single_process(exchange, monkeypatch):
    monkeypatch.setenv("ITX_TEST_EXCHANGE", exchange)      # inherited by the spawned ranks
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    chroms, t, rl, rows, reads = _workload()
    whole = _oracle_partial(chroms, t, rl, rows, reads, 0, len(reads[0]))
    whole["cov"][0] += np.uint32((2 * 0xFFFFFFF0) & 0xFFFFFFFF)
    whole["cnt"][12] += np.uint64((2 * 0xFFFFFFFFFFFFFF00) & 0xFFFFFFFFFFFFFFFF)
    for k in ("cnt", "rep_cnt", "fam_cnt", "cla_cnt", "cov", "cov_uniq"):
        assert np.array_equal(got[k], whole[k]), k
    assert int(whole["cnt"][9]) > 1000


def test_shard_bounds_cover_exactly_once():
    for n in (0, 1, 1023, 1024, 50_000_001):
        for world in (1, 2, 3, 8):
            spans = [idist.shard_bounds(n, r, world, align=1024) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            assert all(lo % 1024 == 0 for lo, _ in spans if lo < n)
