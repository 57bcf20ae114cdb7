#!/usr/bin/env python3
"""Scratch: what hipMalloc / hipFree / hipHostMalloc cost on this box, by size."""
import ctypes as C
import time
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
p = C.c_void_p()
assert hip.hipMalloc(C.byref(p), 1 << 20) == 0
hip.hipFree(p)
for gb in (0.125, 0.5, 1, 2, 8, 32):
    n = int(gb * (1 << 30))
    t0 = time.perf_counter()
    assert hip.hipMalloc(C.byref(p), n) == 0
    t1 = time.perf_counter()
    hip.hipMemset(p, 0, n)
    hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    hip.hipFree(p)
    t3 = time.perf_counter()
    print(f"hipMalloc {gb:7.3f} GiB: {1e3 * (t1 - t0):8.2f} ms ({1e3 * (t1 - t0) / gb:6.1f} ms/GiB)  first memset {1e3 * (t2 - t1):8.2f} ms  hipFree {1e3 * (t3 - t2):8.2f} ms", flush=True)
t0 = time.perf_counter()
ps = []
for i in range(48):
    q = C.c_void_p()
    assert hip.hipMalloc(C.byref(q), int(0.65 * (1 << 30))) == 0
    ps.append(q)
t1 = time.perf_counter()
print(f"48 x 0.65 GiB: {1e3 * (t1 - t0):.1f} ms")
for q in ps:
    hip.hipFree(q)
for mb in (16, 128, 512):
    n = mb << 20
    t0 = time.perf_counter()
    assert hip.hipHostMalloc(C.byref(p), n, 0) == 0
    t1 = time.perf_counter()
    hip.hipHostFree(p)
    print(f"hipHostMalloc {mb} MiB: {1e3 * (t1 - t0):.2f} ms, free {1e3 * (time.perf_counter() - t1):.2f} ms")
# the same after streams exist and a copy has run on one of them
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
sts = []
for i in range(6):
    st = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0
    sts.append(st)
a, b = C.c_void_p(), C.c_void_p()
hip.hipMalloc(C.byref(a), 1 << 26)
hip.hipMalloc(C.byref(b), 1 << 26)
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
for st in sts:
    hip.hipMemcpyAsync(a, b, 1 << 26, 3, st)
hip.hipDeviceSynchronize()
t0 = time.perf_counter()
ps = []
for i in range(48):
    q = C.c_void_p()
    assert hip.hipMalloc(C.byref(q), int(0.8 * (1 << 30))) == 0
    ps.append(q)
print(f"with 6 streams: 48 x 0.8 GiB: {1e3 * (time.perf_counter() - t0):.1f} ms")
t0 = time.perf_counter()
for i in range(4):
    q = C.c_void_p()
    assert hip.hipMalloc(C.byref(q), int(2.6 * (1 << 30))) == 0
    ps.append(q)
print(f"4 x 2.6 GiB: {1e3 * (time.perf_counter() - t0):.1f} ms")
