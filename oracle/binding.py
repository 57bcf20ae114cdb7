"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrcParams(C.Structure):
    _fields_ = [("mapq_min", C.c_uint32), ("min_cov", C.c_float), ("extension", C.c_uint32),
                ("isize_max", C.c_uint32), ("treat_pe_as_se", C.c_int32), ("discard_half_mapped", C.c_int32),
                ("filter_mode", C.c_int32)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("iteres_oracle.c", "iteres_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_table_new.restype = C.c_void_p
        L.orc_table_new.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_table_free.argtypes = [C.c_void_p]
        L.orc_table_add.restype = C.c_int64
        L.orc_table_add.argtypes = [C.c_void_p, C.c_int] + [C.c_uint32] * 7
        L.orc_table_add_many.restype = None
        L.orc_table_add_many.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 9
        L.orc_find.restype = C.c_int64
        L.orc_find.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64]
        L.orc_cov_offsets.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_run.restype = C.c_int
        L.orc_run.argtypes = [C.c_void_p, C.POINTER(OrcParams), C.c_int, C.c_void_p, C.c_size_t] + [C.c_void_p] * 16
        L.orc_hash_string.restype = C.c_uint32
        L.orc_hash_string.argtypes = [C.c_char_p]
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleTable:
    """The reference's hashRmsk (per-chrom binKeeper) + repeat-size lookup, restated."""

    def __init__(self, chrom_size, rep_len, n_fam, n_cla):
        self.chrom_size = np.ascontiguousarray(chrom_size, np.int64)
        self.rep_len = np.ascontiguousarray(rep_len, np.uint32)
        self.n_rep, self.n_fam, self.n_cla = len(self.rep_len), int(n_fam), int(n_cla)
        self._h = lib().orc_table_new(len(self.chrom_size), _p(self.chrom_size), self.n_rep, _p(self.rep_len),
                                      self.n_fam, self.n_cla)
        self.n_rows = 0
        off = np.zeros(self.n_rep + 1, np.uint64)
        lib().orc_cov_offsets(self._h, _p(off))
        self.cov_off = off

    def add_rows(self, chrom, start, end, cons_start, cons_end, rep, fam, cla):
        """Rows in file order. Returns int64 array: stored row index, -1 dropped (chrom unknown), -2 abort."""
        out = np.empty(len(chrom), np.int64)
        u32 = lambda a: np.ascontiguousarray(np.asarray(a).astype(np.int64) & 0xFFFFFFFF, np.uint32)
        arrs = [np.ascontiguousarray(chrom, np.int32)] + [u32(a) for a in (start, end, cons_start, cons_end, rep, fam, cla)]
        lib().orc_table_add_many(self._h, len(out), *[_p(a) for a in arrs], _p(out))
        self.n_rows = int((out >= 0).sum()) + self.n_rows
        return out

    def find(self, chrom, start, end, cap=4096):
        rows = np.empty(cap, np.int64)
        n = lib().orc_find(self._h, int(chrom), int(start), int(end), _p(rows), cap)
        return rows[:min(n, cap)].copy()

    def run(self, params: dict, tid2chrom, tid, pos, tmpend, mapq, flag, mpos=None, isize=None, want_hits=True, skip=None):
        n = len(tid)
        p = OrcParams(int(params.get("mapq_min", 10)), float(np.float32(params.get("min_cov", 0.0001))),
                      int(params.get("extension", 150)), int(params.get("isize_max", 500)),
                      int(bool(params.get("treat_pe_as_se", False))), int(bool(params.get("discard_half_mapped", False))),
                      int(bool(params.get("filter_mode", False))))
        tid2chrom = np.ascontiguousarray(tid2chrom, np.int32)
        tid = np.ascontiguousarray(tid, np.int32)
        pos = np.ascontiguousarray(pos, np.int32)
        tmpend = np.ascontiguousarray(tmpend, np.int32)
        mapq = np.ascontiguousarray(mapq, np.uint8)
        flag = np.ascontiguousarray(flag, np.uint16)
        mpos = np.zeros(n, np.int32) if mpos is None else np.ascontiguousarray(mpos, np.int32)
        isize = np.zeros(n, np.int32) if isize is None else np.ascontiguousarray(isize, np.int32)
        skip = None if skip is None else np.ascontiguousarray(skip, np.uint8)
        res = {
            "hit_row": np.empty(n, np.int64) if want_hits else None,
            "cnt": np.zeros(13, np.uint64),
            "rep_cnt": np.zeros(2 * self.n_rep, np.uint64),
            "fam_cnt": np.zeros(2 * self.n_fam, np.uint64),
            "cla_cnt": np.zeros(2 * self.n_cla, np.uint64),
            "cov": np.zeros(int(self.cov_off[-1]), np.uint32),
            "cov_uniq": np.zeros(int(self.cov_off[-1]), np.uint32),
            "locus_cnt": np.zeros(max(self.n_rows, 1), np.uint32),
        }
        rc = lib().orc_run(self._h, C.byref(p), len(tid2chrom), _p(tid2chrom), n, _p(tid), _p(pos), _p(tmpend),
                           _p(mapq), _p(flag), _p(mpos), _p(isize), _p(skip), _p(res["hit_row"]), _p(res["cnt"]),
                           _p(res["rep_cnt"]), _p(res["fam_cnt"]), _p(res["cla_cnt"]), _p(res["cov"]),
                           _p(res["cov_uniq"]), _p(res["locus_cnt"]))
        assert rc == 0
        return res

    def close(self):
        if self._h:
            lib().orc_table_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def hash_string(s: str) -> int:
    return int(lib().orc_hash_string(s.encode()))
