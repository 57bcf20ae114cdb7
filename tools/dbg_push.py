import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from iteres_amd import engine as eng, synth
chroms = [("c1", 3_000_000)]
r = synth.make_reads(101, chroms, 20000, read_len=(30, 60))
synth.write_bam("/tmp/dbg.bam", r, with_seq=True)
comp = open("/tmp/dbg.bam", "rb").read()
blocks = eng.index_bgzf(comp)
print("blocks", len(blocks), flush=True)
L = eng.load()
L.itx_bamwin_push_begin.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
L.itx_bamwin_push_end.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_size_t)]
h = eng.Inflater()
cbuf = np.zeros(len(comp) + 64, np.uint8); cbuf[:len(comp)] = np.frombuffer(comp, np.uint8)
status = np.zeros(len(blocks), np.uint8)
n_new = C.c_size_t(0)
print("push begin", flush=True)
rc = L.itx_bamwin_push_begin(h._h, 0, 0, eng._p(cbuf), len(comp), eng._p(blocks), len(blocks))
print("begin rc", rc, L.itx_last_error(), flush=True)
rc = L.itx_bamwin_push_end(h._h, 0, eng._p(status), C.byref(n_new))
print("end rc", rc, n_new.value, status.sum(), flush=True)
