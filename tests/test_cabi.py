"""CPU-side checks of the drop-in boundary: the shared library loads and exports exactly the entry points
include/iteres_amd.h declares; argument validation that needs no GPU behaves as documented."""
import ctypes as C
import os
import re

import pytest

from iteres_amd import build, engine as eng

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_lib()
    return eng.load()


def _declared():
    src = open(os.path.join(ROOT, "include", "iteres_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(itx_[a-z0-9_]+)\s*\(", src)))


def test_exports_every_declared_symbol(lib):
    names = _declared()
    assert set(names) == set(eng.EXPORTS)
    for n in names:
        assert getattr(lib, n) is not None


def test_abi_version_and_error_paths(lib):
    assert lib.itx_abi_version() == 1005
    assert lib.itx_table_get_info(None, None) == -1                 # ITX_E_ARG
    assert b"null" in lib.itx_last_error()
    h = C.c_void_p()
    bad = C.c_size_t(0)
    assert lib.itx_table_create(None, 5, None, 1, None, 0, 0, 0, 0, C.byref(h), C.byref(bad)) == -1
    assert lib.itx_engine_create(None, None, 0, C.byref(h)) == -1
    assert lib.itx_engine_finish(None, None) == -1


def test_table_validation_needs_no_gpu(lib):
    """A row past its chromosome end is what binKeeperAdd aborts on (cuskent/binRange.c:176-178): rejected before
    any device work, with the offending row reported."""
    import numpy as np
    rows = eng.make_rows([0, 0], [10, 90], [20, 120], [0, 0], [5, 5], [0, 0], [0, 0], [0, 0])
    cs = np.array([100], np.int64)
    rl = np.array([50], np.uint32)
    h = C.c_void_p()
    bad = C.c_size_t(99)
    rc = lib.itx_table_create(rows.ctypes.data_as(C.c_void_p), 2, cs.ctypes.data_as(C.c_void_p), 1, rl.ctypes.data_as(C.c_void_p),
                              1, 1, 1, 0, C.byref(h), C.byref(bad))
    assert rc == -2 and bad.value == 1


def test_no_cpu_fallback_without_gpu(lib):
    """Without a GPU the product must fail loudly, never compute on the host."""
    import numpy as np
    if lib.itx_device_count() > 0:
        pytest.skip("a GPU is present")
    rows = eng.make_rows([0], [10], [20], [0], [5], [0], [0], [0])
    with pytest.raises(eng.ItxError):
        eng.Table(rows, [100], np.array([50], np.uint32), 1, 1)
