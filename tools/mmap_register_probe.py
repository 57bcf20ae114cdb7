#!/usr/bin/env python3
"""Scratch probe: can the device read a file straight out of the page cache? mmap a file read-only, hipHostRegister the
mapping (plain, then with hipHostRegisterReadOnly), hipMemcpy it to the device and back, compare; time every step next to
pread into a page-locked buffer.   python tools/mmap_register_probe.py [MB=1024]"""
import ctypes as C
import mmap
import os
import sys
import time

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = mb << 20
path = "/tmp/itx_probe.bin"
if not os.path.exists(path) or os.path.getsize(path) != n:
    with open(path, "wb") as f:
        blk = os.urandom(1 << 20)
        for _ in range(mb):
            f.write(blk)
hip = C.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = C.c_char_p


def chk(rc, what):
    print(f"{what}: rc={rc} {hip.hipGetErrorString(rc).decode() if rc else 'ok'}", flush=True)
    return rc == 0


hip.hipSetDevice(0)
dev = C.c_void_p()
chk(hip.hipMalloc(C.byref(dev), C.c_size_t(n)), "hipMalloc")
fd = os.open(path, os.O_RDONLY)
# warm the page cache
t0 = time.perf_counter()
while os.read(fd, 1 << 24):
    pass
print(f"read through once: {time.perf_counter() - t0:.3f} s")
libc = C.CDLL("libc.so.6", use_errno=True)
libc.mmap.restype = C.c_void_p
libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
for prot, flags, name in ((mmap.PROT_READ, mmap.MAP_SHARED, "PROT_READ MAP_SHARED"), (mmap.PROT_READ, mmap.MAP_PRIVATE, "PROT_READ MAP_PRIVATE")):
    p = libc.mmap(None, n, prot, flags, fd, 0)
    if p in (None, C.c_void_p(-1).value):
        print(name, "mmap failed", C.get_errno())
        continue
    for hflag, hname in ((0, "default"), (8, "hipHostRegisterReadOnly"), (2, "mapped"), (10, "mapped|readonly")):
        t0 = time.perf_counter()
        rc = hip.hipHostRegister(C.c_void_p(p), C.c_size_t(n), C.c_uint(hflag))
        dt = time.perf_counter() - t0
        ok = chk(rc, f"{name}: hipHostRegister({hname}) {dt:.3f} s ({n / dt / 1e9:.1f} GB/s)")
        if ok:
            t0 = time.perf_counter()
            rc = hip.hipMemcpy(dev, C.c_void_p(p), C.c_size_t(n), C.c_int(1))
            dt = time.perf_counter() - t0
            chk(rc, f"   hipMemcpy H2D from the registered mapping {dt:.3f} s ({n / dt / 1e9:.1f} GB/s)")
            back = (C.c_char * (1 << 20))()
            hip.hipMemcpy(back, C.c_void_p(dev.value + n - (1 << 20)), C.c_size_t(1 << 20), C.c_int(2))
            with open(path, "rb") as f:
                f.seek(n - (1 << 20))
                print("   last MiB equal:", f.read(1 << 20) == bytes(back), flush=True)
            t0 = time.perf_counter()
            chk(hip.hipHostUnregister(C.c_void_p(p)), "   hipHostUnregister")
            print(f"   unregister {time.perf_counter() - t0:.3f} s")
            break
    libc.munmap(C.c_void_p(p), C.c_size_t(n))
# the route in use: pread into a page-locked buffer, one thread
host = C.c_void_p()
t0 = time.perf_counter()
chk(hip.hipHostMalloc(C.byref(host), C.c_size_t(n), C.c_uint(0)), "hipHostMalloc")
print(f"page-locking {mb} MB: {time.perf_counter() - t0:.3f} s")
libc.pread.restype = C.c_ssize_t
libc.pread.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_long]
t0 = time.perf_counter()
at = 0
while at < n:
    k = libc.pread(fd, C.c_void_p(host.value + at), n - at, at)
    if k <= 0:
        break
    at += k
dt = time.perf_counter() - t0
print(f"pread into the page-locked buffer, one thread: {dt:.3f} s ({n / dt / 1e9:.1f} GB/s)")
t0 = time.perf_counter()
hip.hipMemcpy(dev, host, C.c_size_t(n), C.c_int(1))
dt = time.perf_counter() - t0
print(f"hipMemcpy H2D from it: {dt:.3f} s ({n / dt / 1e9:.1f} GB/s)")
# pageable source straight from a plain (unregistered) mapping
p = libc.mmap(None, n, mmap.PROT_READ, mmap.MAP_SHARED, fd, 0)
t0 = time.perf_counter()
rc = hip.hipMemcpy(dev, C.c_void_p(p), C.c_size_t(n), C.c_int(1))
dt = time.perf_counter() - t0
chk(rc, f"hipMemcpy H2D straight from an unregistered mapping {dt:.3f} s ({n / dt / 1e9:.1f} GB/s)")
