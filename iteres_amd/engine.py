"""ctypes binding of include/iteres_amd.h (libiteres_amd.so) for the test-suite, bench.py and smoke().

This is plumbing above the C ABI, not a second implementation: every call lands in the HIP engine.
If the shared library is missing or no GPU is usable the calls raise — there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ITX_LIB", os.path.join(HERE, "libiteres_amd.so"))   # ITX_LIB: timing-only experiment builds

MODE_STAT, MODE_FILTER = 0, 1
F5_NOLOOKUP = 0x20        # include/iteres_amd.h ITX_F5_NOLOOKUP
ACCUM_DEFAULT, ACCUM_ATOMIC, ACCUM_PARTITION = 0, 1, 2

EXPORTS = [
    "itx_last_error", "itx_abi_version", "itx_device_count", "itx_table_create", "itx_table_destroy",
    "itx_table_get_info", "itx_table_cov_offsets", "itx_engine_create", "itx_engine_destroy", "itx_engine_set_tidmap",
    "itx_engine_staging", "itx_engine_submit_slot", "itx_engine_classify_slot", "itx_engine_wait_slot", "itx_engine_submit_device",
    "itx_engine_classify_device", "itx_engine_first_hit_slot", "itx_engine_first_hit_device", "itx_engine_sync", "itx_engine_reset", "itx_engine_finish", "itx_engine_get_stats",
    "itx_engine_partial_size", "itx_engine_export_partial", "itx_engine_finish_partial",
    "itx_inflater_create", "itx_inflater_destroy", "itx_inflate_bgzf", "itx_inflater_last_ms", "itx_pinned_alloc", "itx_pinned_free",
    "itx_bamwin_push", "itx_bamwin_push_begin", "itx_bamwin_push_copied", "itx_bamwin_push_end", "itx_bamwin_patch", "itx_bamwin_truncate", "itx_bamwin_carry", "itx_bamwin_avail", "itx_bamwin_peek", "itx_bamwin_skip",
    "itx_bamwin_parse", "itx_bamwin_fetch", "itx_bamwin_bytes", "itx_bamwin_tids", "itx_bamwin_device_batch",
    "itx_engine_submit_device_own", "itx_engine_wait_own",
    "itx_engine_partial_buffers", "itx_inflater_reserve", "itx_inflater_last_resolve_all_ms", "itx_timing_report", "itx_xaveto_create", "itx_xaveto_destroy", "itx_xaveto_set_tidmap", "itx_xaveto_hits", "itx_xaveto_stream", "itx_bamwin_xa_veto", "itx_comm_create", "itx_comm_destroy", "itx_comm_reduce_sum",
    "itx_backlog_create", "itx_backlog_destroy", "itx_backlog_room", "itx_backlog_append", "itx_backlog_batch", "itx_dedup_create", "itx_dedup_destroy", "itx_dedup_set_tidmap", "itx_dedup_run", "itx_dedup_counts", "itx_bamwin_dedup",
]


class ItxError(RuntimeError):
    pass


class Row(C.Structure):
    _fields_ = [("chrom", C.c_int32), ("start", C.c_uint32), ("end", C.c_uint32), ("cons_start", C.c_uint32),
                ("cons_end", C.c_uint32), ("rep", C.c_uint32), ("fam", C.c_uint32), ("cla", C.c_uint32)]


ROW_DTYPE = np.dtype([("chrom", "<i4"), ("start", "<u4"), ("end", "<u4"), ("cons_start", "<u4"), ("cons_end", "<u4"),
                      ("rep", "<u4"), ("fam", "<u4"), ("cla", "<u4")])
assert ROW_DTYPE.itemsize == C.sizeof(Row) == 32


class TableInfo(C.Structure):
    _fields_ = [("n_rows", C.c_uint64), ("n_rep", C.c_uint64), ("n_fam", C.c_uint64), ("n_cla", C.c_uint64),
                ("cov_len", C.c_uint64), ("n_units", C.c_uint64), ("n_slots", C.c_uint64), ("table_bytes", C.c_uint64),
                ("n_chrom", C.c_int32), ("bin_shift", C.c_int32), ("device", C.c_int32), ("reserved", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("mapq_min", C.c_uint32), ("min_cov", C.c_float), ("extension", C.c_uint32), ("isize_max", C.c_uint32),
                ("treat_pe_as_se", C.c_int32), ("discard_half_mapped", C.c_int32), ("mode", C.c_int32), ("accum", C.c_int32)]


class Batch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("tid", "pos", "tmpend", "mapq", "flag5", "mpos", "isize")]


class Staging(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("tid", "pos", "tmpend", "mapq", "flag5", "mpos", "isize", "hit_row")] + [("capacity", C.c_size_t)]


class Result(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("cnt", "rep_cnt", "fam_cnt", "cla_cnt", "cov", "cov_uniq", "locus_cnt")]


class Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("records", C.c_uint64), ("hits", C.c_uint64), ("stage_ms", C.c_double * 5),
                ("submits", C.c_uint64), ("keys", C.c_uint64)]


_lib = None


def load():
    """Loads the in-tree shared library; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ItxError(f"{LIB_PATH} is missing: build it with `python -m iteres_amd.build` (hipcc, gfx950)")
    L = C.CDLL(LIB_PATH)
    L.itx_last_error.restype = C.c_char_p
    L.itx_table_create.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                   C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.itx_table_destroy.argtypes = [C.c_void_p]
    L.itx_table_destroy.restype = None
    L.itx_table_get_info.argtypes = [C.c_void_p, C.POINTER(TableInfo)]
    L.itx_table_cov_offsets.argtypes = [C.c_void_p, C.c_void_p]
    L.itx_engine_create.argtypes = [C.c_void_p, C.POINTER(Params), C.c_size_t, C.POINTER(C.c_void_p)]
    L.itx_engine_partial_size.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.itx_engine_export_partial.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.itx_engine_finish_partial.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Result)]
    L.itx_engine_partial_buffers.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.itx_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    L.itx_comm_destroy.argtypes = [C.c_void_p]
    L.itx_comm_destroy.restype = None
    L.itx_comm_reduce_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    L.itx_engine_destroy.argtypes = [C.c_void_p]
    L.itx_engine_destroy.restype = None
    L.itx_engine_set_tidmap.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.itx_engine_staging.argtypes = [C.c_void_p, C.c_int, C.POINTER(Staging)]
    L.itx_engine_submit_slot.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int]
    L.itx_engine_classify_slot.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_int]
    L.itx_engine_first_hit_slot.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    L.itx_engine_first_hit_device.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_size_t, C.c_void_p, C.c_void_p]
    L.itx_engine_wait_slot.argtypes = [C.c_void_p, C.c_int]
    L.itx_engine_submit_device.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_size_t, C.c_void_p, C.c_void_p]
    L.itx_engine_classify_device.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_size_t, C.c_void_p, C.c_void_p]
    L.itx_engine_sync.argtypes = [C.c_void_p]
    L.itx_engine_reset.argtypes = [C.c_void_p]
    L.itx_engine_finish.argtypes = [C.c_void_p, C.POINTER(Result)]
    L.itx_engine_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.itx_inflater_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.itx_inflater_destroy.argtypes = [C.c_void_p]
    L.itx_inflater_destroy.restype = None
    L.itx_inflate_bgzf.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    L.itx_inflater_last_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.itx_bamwin_push.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
    L.itx_bamwin_patch.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t]
    L.itx_bamwin_truncate.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    L.itx_bamwin_carry.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.itx_bamwin_avail.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
    L.itx_bamwin_peek.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t]
    L.itx_bamwin_skip.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    L.itx_bamwin_parse.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    L.itx_bamwin_fetch.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(Staging), C.c_size_t, C.c_void_p, C.c_void_p]
    L.itx_bamwin_bytes.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.itx_dedup_create.argtypes = [C.c_int, C.c_void_p, C.c_int, C.POINTER(Params), C.c_size_t, C.POINTER(C.c_void_p)]
    L.itx_dedup_destroy.argtypes = [C.c_void_p]
    L.itx_dedup_destroy.restype = None
    L.itx_dedup_set_tidmap.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.itx_dedup_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.itx_dedup_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.itx_pinned_alloc.argtypes = [C.c_size_t]
    L.itx_pinned_alloc.restype = C.c_void_p
    L.itx_pinned_free.argtypes = [C.c_void_p]
    L.itx_pinned_free.restype = None
    _lib = L
    return L


def _chk(rc, what):
    if rc != 0:
        raise ItxError(f"{what}: rc={rc}: {load().itx_last_error().decode(errors='replace')}")


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def flag5(bamflag):
    f = np.asarray(bamflag).astype(np.uint32)
    return (((f & 0x1) != 0) * 1 + ((f & 0x4) != 0) * 2 + ((f & 0x8) != 0) * 4 + ((f & 0x10) != 0) * 8 + ((f & 0x40) != 0) * 16).astype(np.uint8)


def make_rows(chrom, start, end, cons_start, cons_end, rep, fam, cla):
    n = len(chrom)
    rows = np.empty(n, ROW_DTYPE)
    u = lambda a: (np.asarray(a).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)
    rows["chrom"] = np.asarray(chrom, np.int32)
    for k, v in (("start", start), ("end", end), ("cons_start", cons_start), ("cons_end", cons_end), ("rep", rep), ("fam", fam), ("cla", cla)):
        rows[k] = u(v)
    return rows


class Table:
    def __init__(self, rows, chrom_size, rep_len, n_fam, n_cla, device=0):
        L = load()
        self.rows = np.ascontiguousarray(rows, ROW_DTYPE)
        cs = np.ascontiguousarray(chrom_size, np.int64)
        rl = np.ascontiguousarray(rep_len, np.uint32)
        h = C.c_void_p()
        bad = C.c_size_t(0)
        rc = L.itx_table_create(_p(self.rows), len(self.rows), _p(cs), len(cs), _p(rl), len(rl), int(n_fam), int(n_cla), int(device),
                                C.byref(h), C.byref(bad))
        self.bad_row = bad.value
        _chk(rc, "itx_table_create")
        self._h = h
        self.info = TableInfo()
        _chk(L.itx_table_get_info(self._h, C.byref(self.info)), "itx_table_get_info")
        self.cov_off = np.zeros(len(rl) + 1, np.uint64)
        _chk(L.itx_table_cov_offsets(self._h, _p(self.cov_off)), "itx_table_cov_offsets")
        self.n_rep, self.n_fam, self.n_cla, self.n_rows = len(rl), int(n_fam), int(n_cla), len(self.rows)
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            load().itx_table_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    def __init__(self, table: Table, params: dict | None = None, batch_capacity: int = 1 << 20):
        L = load()
        q = dict(mapq_min=10, min_cov=0.0001, extension=150, isize_max=500, treat_pe_as_se=False, discard_half_mapped=False,
                 filter_mode=False, accum=ACCUM_DEFAULT)
        q.update(params or {})
        self.params = Params(int(q["mapq_min"]), float(np.float32(q["min_cov"])), int(q["extension"]), int(q["isize_max"]),
                             int(bool(q["treat_pe_as_se"])), int(bool(q["discard_half_mapped"])),
                             MODE_FILTER if q.get("filter_mode") else MODE_STAT, int(q["accum"]))
        self.table = table
        h = C.c_void_p()
        _chk(L.itx_engine_create(table._h, C.byref(self.params), int(batch_capacity), C.byref(h)), "itx_engine_create")
        self._h = h
        self.capacity = int(batch_capacity)

    def set_tidmap(self, tid2chrom):
        a = np.ascontiguousarray(tid2chrom, np.int32)
        _chk(load().itx_engine_set_tidmap(self._h, _p(a), len(a)), "itx_engine_set_tidmap")

    def staging(self, slot):
        st = Staging()
        _chk(load().itx_engine_staging(self._h, slot, C.byref(st)), "itx_engine_staging")
        n = st.capacity

        def view(ptr, ctype, dt):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).view(dt)
        return {"tid": view(st.tid, C.c_int32, np.int32), "pos": view(st.pos, C.c_int32, np.int32),
                "tmpend": view(st.tmpend, C.c_int32, np.int32), "mapq": view(st.mapq, C.c_uint8, np.uint8),
                "flag5": view(st.flag5, C.c_uint8, np.uint8), "mpos": view(st.mpos, C.c_int32, np.int32),
                "isize": view(st.isize, C.c_int32, np.int32), "hit_row": view(st.hit_row, C.c_int32, np.int32)}

    def submit_host(self, tid, pos, tmpend, mapq, flag5_, mpos=None, isize=None, want_hits=False, veto=None):
        """Streams host arrays through the pinned double buffers (slot ping-pong). Returns hit rows if asked.
        veto(offset, hit_rows) -> bool mask: records to mark ITX_F5_NOLOOKUP after a classify-only pass."""
        L = load()
        n = len(tid)
        hits = np.empty(n, np.int32) if want_hits else None
        paired = mpos is not None and isize is not None
        bufs = [self.staging(0), self.staging(1)]
        pending = [None, None]
        off, s = 0, 0
        while off < n or any(p is not None for p in pending):
            if pending[s] is not None:
                _chk(L.itx_engine_wait_slot(self._h, s), "itx_engine_wait_slot")
                a, b = pending[s]
                if want_hits:
                    hits[a:b] = bufs[s]["hit_row"][: b - a]
                pending[s] = None
            if off < n:
                m = min(self.capacity, n - off)
                sl = slice(off, off + m)
                bufs[s]["tid"][:m] = tid[sl]; bufs[s]["pos"][:m] = pos[sl]; bufs[s]["tmpend"][:m] = tmpend[sl]
                bufs[s]["mapq"][:m] = mapq[sl]; bufs[s]["flag5"][:m] = flag5_[sl]
                if paired:
                    bufs[s]["mpos"][:m] = mpos[sl]; bufs[s]["isize"][:m] = isize[sl]
                if veto is not None:
                    # what a caller with an XA-style veto does: classify, look at the chosen rows, mark, then count
                    _chk(L.itx_engine_classify_slot(self._h, s, m, int(paired)), "itx_engine_classify_slot")
                    _chk(L.itx_engine_wait_slot(self._h, s), "itx_engine_wait_slot")
                    mask = np.asarray(veto(off, bufs[s]["hit_row"][:m].copy()), bool)
                    bufs[s]["flag5"][:m] |= mask.astype(np.uint8) * np.uint8(F5_NOLOOKUP)
                _chk(L.itx_engine_submit_slot(self._h, s, m, int(paired), int(want_hits)), "itx_engine_submit_slot")
                pending[s] = (off, off + m)
                off += m
            s ^= 1
        return hits

    def first_hits_host(self, tid, start, end):
        """First row in binKeeperFind's order for every plain interval (itx_engine_first_hit_slot), through the slots."""
        L = load()
        n = len(tid)
        hits = np.empty(n, np.int32)
        bufs = [self.staging(0), self.staging(1)]
        pending = [None, None]
        off, s = 0, 0
        while off < n or any(p is not None for p in pending):
            if pending[s] is not None:
                _chk(L.itx_engine_wait_slot(self._h, s), "itx_engine_wait_slot")
                a, b = pending[s]
                hits[a:b] = bufs[s]["hit_row"][: b - a]
                pending[s] = None
            if off < n:
                m = min(self.capacity, n - off)
                sl = slice(off, off + m)
                bufs[s]["tid"][:m] = tid[sl]; bufs[s]["pos"][:m] = start[sl]; bufs[s]["tmpend"][:m] = end[sl]
                bufs[s]["mapq"][:m] = 0; bufs[s]["flag5"][:m] = 0
                _chk(L.itx_engine_first_hit_slot(self._h, s, m), "itx_engine_first_hit_slot")
                pending[s] = (off, off + m)
                off += m
            s ^= 1
        return hits

    def submit_device(self, ptrs: dict, n: int, hit_ptr=None, stream=None, classify_only=False):
        b = Batch(*(ptrs.get(k) for k in ("tid", "pos", "tmpend", "mapq", "flag5", "mpos", "isize")))
        fn = load().itx_engine_classify_device if classify_only else load().itx_engine_submit_device
        _chk(fn(self._h, C.byref(b), int(n), hit_ptr, stream), "itx_engine_submit_device")

    def sync(self):
        _chk(load().itx_engine_sync(self._h), "itx_engine_sync")

    def reset(self):
        _chk(load().itx_engine_reset(self._h), "itx_engine_reset")

    def _result_arrays(self):
        t = self.table
        res = {"cnt": np.zeros(13, np.uint64), "rep_cnt": np.zeros(2 * t.n_rep, np.uint64), "fam_cnt": np.zeros(2 * t.n_fam, np.uint64),
               "cla_cnt": np.zeros(2 * t.n_cla, np.uint64), "cov": np.zeros(int(t.info.cov_len), np.uint32),
               "cov_uniq": np.zeros(int(t.info.cov_len), np.uint32), "locus_cnt": np.zeros(max(t.n_rows, 1), np.uint32)}
        r = Result(*(_p(res[k]) for k in ("cnt", "rep_cnt", "fam_cnt", "cla_cnt", "cov", "cov_uniq", "locus_cnt")))
        return res, r

    def finish(self):
        res, r = self._result_arrays()
        _chk(load().itx_engine_finish(self._h, C.byref(r)), "itx_engine_finish")
        return res

    def partial_size(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        _chk(load().itx_engine_partial_size(self._h, C.byref(a), C.byref(b)), "itx_engine_partial_size")
        return int(a.value), int(b.value)

    def export_partial(self, u64_ptr, u32_ptr, stream=None):
        _chk(load().itx_engine_export_partial(self._h, u64_ptr, u32_ptr, stream), "itx_engine_export_partial")

    def finish_partial(self, u64_ptr, u32_ptr):
        res, r = self._result_arrays()
        _chk(load().itx_engine_finish_partial(self._h, u64_ptr, u32_ptr, C.byref(r)), "itx_engine_finish_partial")
        return res

    def stats(self):
        s = Stats()
        _chk(load().itx_engine_get_stats(self._h, C.byref(s)), "itx_engine_get_stats")
        return {"kernel_ms": s.kernel_ms, "records": int(s.records), "hits": int(s.hits), "stage_ms": list(s.stage_ms),
                "submits": int(s.submits), "keys": int(s.keys)}

    def close(self):
        if getattr(self, "_h", None):
            load().itx_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


BGZF_BLOCK = np.dtype([("coff", np.uint32), ("csize", np.uint32), ("uoff", np.uint32), ("usize", np.uint32)])


def index_bgzf(buf) -> np.ndarray:
    """The complete BGZF blocks of a byte buffer (what the host reader's indexer finds: header check, BSIZE, ISIZE)."""
    b = np.frombuffer(buf, np.uint8)
    out, off, uoff = [], 0, 0
    while off + 18 <= len(b):
        h = b[off:off + 18]
        if not (h[0] == 31 and h[1] == 139 and h[2] == 8 and (h[3] & 4) and h[12] == 66 and h[13] == 67):
            break
        bsize = int(h[16]) + (int(h[17]) << 8) + 1
        if off + bsize > len(b):
            break
        usize = int.from_bytes(bytes(b[off + bsize - 4:off + bsize]), "little")
        out.append((off, bsize, uoff, usize))
        off += bsize
        uoff += usize
    return np.array(out, BGZF_BLOCK)


class Inflater:
    """itx_inflate_bgzf: the BGZF blocks of a compressed chunk, one wavefront each (include/iteres_amd.h)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _chk(load().itx_inflater_create(device, C.byref(self._h)), "itx_inflater_create")

    def inflate(self, comp: bytes, blocks: np.ndarray | None = None):
        blocks = index_bgzf(comp) if blocks is None else np.ascontiguousarray(blocks, BGZF_BLOCK)
        cbuf = np.zeros(len(comp) + 16, np.uint8)
        cbuf[:len(comp)] = np.frombuffer(comp, np.uint8)
        total = int(blocks["usize"].astype(np.uint64).sum())
        out = np.zeros(total + 16, np.uint8)
        status = np.full(max(len(blocks), 1), 255, np.uint8)
        _chk(load().itx_inflate_bgzf(self._h, _p(cbuf), len(comp), _p(blocks), len(blocks), _p(out), total, _p(status)), "itx_inflate_bgzf")
        return out[:total], status[:len(blocks)]

    def close(self):
        if self._h:
            load().itx_inflater_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Dedup:
    """-R on the device (include/iteres_amd.h itx_dedup_*, replacing generic.c:907-919): `run` takes the next records of the
    stream as torch DEVICE tensors (int32 tid / pos / tmpend, uint8 mapq / flag5, optional int32 mpos / isize) and ORs
    ITX_F5_NOLOOKUP into flag5 for the records the reference would `continue` over."""

    def __init__(self, chrom_size, params: dict | None = None, first_cells: int = 1 << 16, device: int = 0):
        L = load()
        q = dict(mapq_min=10, min_cov=1e-4, extension=150, isize_max=500, treat_pe_as_se=False, discard_half_mapped=False)
        q.update(params or {})
        p = Params(int(q["mapq_min"]), float(q["min_cov"]), int(q["extension"]), int(q["isize_max"]), int(bool(q["treat_pe_as_se"])),
                   int(bool(q["discard_half_mapped"])), MODE_STAT, ACCUM_DEFAULT)
        cs = np.ascontiguousarray(chrom_size, np.int64)
        self._h = C.c_void_p()
        _chk(L.itx_dedup_create(device, _p(cs), len(cs), C.byref(p), first_cells, C.byref(self._h)), "itx_dedup_create")

    def set_tidmap(self, tid2chrom, tid2name):
        a = np.ascontiguousarray(tid2chrom, np.int32)
        b = np.ascontiguousarray(tid2name, np.uint32)
        _chk(load().itx_dedup_set_tidmap(self._h, _p(a), _p(b), len(a)), "itx_dedup_set_tidmap")

    def run(self, tid, pos, tmpend, mapq, flag5, mpos=None, isize=None):
        n = int(tid.numel())
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        _chk(load().itx_dedup_run(self._h, ptr(tid), ptr(pos), ptr(tmpend), ptr(mapq), ptr(flag5), ptr(mpos), ptr(isize), n), "itx_dedup_run")

    def counts(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _chk(load().itx_dedup_counts(self._h, C.byref(a), C.byref(b), C.byref(c)), "itx_dedup_counts")
        return {"dup_unique": a.value, "dropped": b.value, "keys": c.value}

    def close(self):
        if self._h:
            load().itx_dedup_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
