#!/usr/bin/env python3
"""Scratch probe: what page-locking a 384 MB chunk buffer costs — hipHostMalloc against mmap + MADV_HUGEPAGE + first touch +
hipHostRegister, and against hipHostRegister of plain malloc'ed memory.   python tools/pin_probe.py [MB=384]"""
import ctypes as C
import mmap
import sys
import time

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 384
n = mb << 20
hip = C.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = C.c_char_p
libc = C.CDLL("libc.so.6", use_errno=True)
libc.mmap.restype = C.c_void_p
libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
libc.memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipSetDevice(0)
dev = C.c_void_p()
hip.hipMalloc(C.byref(dev), C.c_size_t(n))
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| defrag:", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())
for rep in range(3):
    h = C.c_void_p()
    t0 = time.perf_counter()
    rc = hip.hipHostMalloc(C.byref(h), C.c_size_t(n), C.c_uint(0))
    t1 = time.perf_counter()
    hip.hipMemcpy(dev, h, C.c_size_t(n), C.c_int(1))
    t2 = time.perf_counter()
    hip.hipHostFree(h)
    t3 = time.perf_counter()
    print(f"hipHostMalloc {mb} MB: rc {rc} {t1 - t0:.3f} s, H2D {n / (t2 - t1) / 1e9:.1f} GB/s, free {t3 - t2:.3f} s", flush=True)
MADV_HUGEPAGE = 14
for huge in (0, 1):
    for rep in range(3):
        t0 = time.perf_counter()
        p = libc.mmap(None, n + (2 << 20), mmap.PROT_READ | mmap.PROT_WRITE, mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, -1, 0)
        a = (p + (2 << 20) - 1) & ~((2 << 20) - 1)
        if huge:
            libc.madvise(C.c_void_p(a), n, MADV_HUGEPAGE)
        t1 = time.perf_counter()
        libc.memset(C.c_void_p(a), 0, n)                       # first touch
        t2 = time.perf_counter()
        rc = hip.hipHostRegister(C.c_void_p(a), C.c_size_t(n), C.c_uint(0))
        t3 = time.perf_counter()
        hip.hipMemcpy(dev, C.c_void_p(a), C.c_size_t(n), C.c_int(1))
        t4 = time.perf_counter()
        hip.hipMemcpy(dev, C.c_void_p(a), C.c_size_t(n), C.c_int(1))
        t5 = time.perf_counter()
        hip.hipHostUnregister(C.c_void_p(a))
        t6 = time.perf_counter()
        print(f"mmap{' + MADV_HUGEPAGE' if huge else ''}: touch {t2 - t1:.3f} s, hipHostRegister rc {rc} {t3 - t2:.3f} s, first H2D {n / (t4 - t3) / 1e9:.1f} GB/s, second {n / (t5 - t4) / 1e9:.1f} GB/s, unregister {t6 - t5:.3f} s", flush=True)
