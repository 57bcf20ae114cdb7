"""The wave's DEFLATE decoder (iteres_amd/csrc/itx_inflate_core.h — the kernel that replaces the reference's per-block zlib
inflate, cussamtools/bgzf.c:367-397) built for the host with a one-lane wave (tests/inflate_host.cpp) and checked against
zlib: every block type, every compression level and strategy, matches at every distance up to 32 KiB, self-overlapping
matches, blocks at odd offsets in the compressed buffer and in the output, and damaged input (must fail cleanly or —
when it decodes — agree with what zlib makes of the same bytes)."""
import ctypes as C
import os
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# both input forms of pass 1: one word of look-ahead (-DITXI_SIMPLE_IN) and the 16-byte read-ahead FIFO (the build's); and
# the loop taking one or three literal/length codes per turn instead of the build's two (ITXI_LITS); the experiment build with
# one 16-bit symbol table (ITXI_SYM16)
INPUT_FORMS = [["-DITXI_SIMPLE_IN"], [], ["-DITXI_LITS=1"], ["-DITXI_LITS=3"], ["-DITXI_SYM16"]]


FORM_IDS = ["word_ahead", "fifo", "one_code_per_turn", "three_codes_per_turn", "symbols_16_bit"]


@pytest.fixture(scope="module", params=INPUT_FORMS, ids=FORM_IDS)
def lib(request, tmp_path_factory):
    so = str(tmp_path_factory.mktemp("inflate") / "libinflate_host.so")
    subprocess.check_call(["g++", "-O2", "-g", "-shared", "-fPIC", "-Wno-unknown-pragmas"] + request.param + ["-o", so, os.path.join(ROOT, "tests", "inflate_host.cpp")])
    L = C.CDLL(so)
    L.itx_inflate_host.restype = C.c_int
    L.itx_inflate_host.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    return L


def raw_deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15, memlevel=8):
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    return co.compress(data) + co.flush()


def run(lib, comp, usize, at=0, g0=0, pad_out=0):
    """comp placed `at` bytes into a word buffer; output written at offset g0 of a buffer with guard bytes around"""
    buf = np.zeros((at + len(comp) + 16 + 3) // 4 + 2, np.uint32)
    buf.view(np.uint8)[at:at + len(comp)] = np.frombuffer(comp, np.uint8)
    out = np.full(g0 + usize + 64 + pad_out, 0xA5, np.uint8)
    rc = lib.itx_inflate_host(buf.ctypes.data, at, at + len(comp), out.ctypes.data, g0, usize, None, None)
    assert (out[:g0] == 0xA5).all() and (out[g0 + usize:] == 0xA5).all(), "wrote outside its block"
    return rc, out[g0:g0 + usize].tobytes()


def corpus(rng):
    yield b""
    yield b"a"
    yield b"abc" * 5
    yield bytes(70000 % 65280)
    yield bytes(rng.integers(0, 256, 65280, dtype=np.uint8))                      # incompressible: stored blocks at level >= 1
    yield bytes(rng.integers(0, 4, 65280, dtype=np.uint8))                        # tiny alphabet: short codes, long matches
    yield (b"ACGT" * 7 + b"N") * 2200
    yield bytes(rng.integers(65, 91, 40000, dtype=np.uint8)) + bytes(25000)
    # BAM-like records: little-endian ints, names, 4-bit sequence, quality runs
    rec = bytearray()
    for i in range(400):
        rec += int(rng.integers(0, 1 << 28)).to_bytes(4, "little") * 3 + f"read{i}".encode() + b"\0" + bytes(rng.integers(0, 256, 50, dtype=np.uint8)) + bytes([40]) * 100
    yield bytes(rec)
    # far matches: a random kilobyte repeated at distances up to the 32 KiB window
    base = bytes(rng.integers(0, 256, 1000, dtype=np.uint8))
    for gap in (7000, 7800, 7900, 8200, 20000, 31700):
        yield base + bytes(rng.integers(0, 256, gap - 1000, dtype=np.uint8)) + base + bytes(rng.integers(0, 256, 500, dtype=np.uint8)) + base[:300]
    # self-overlapping matches of every small distance
    for d in (1, 2, 3, 5, 7, 31, 32, 33, 63, 64, 65, 100, 257, 258, 259):
        yield bytes(rng.integers(0, 256, d, dtype=np.uint8)) * (3000 // d + 2)


def test_against_zlib_every_level_and_strategy(lib):
    rng = np.random.default_rng(5)
    n = 0
    for data in corpus(rng):
        data = data[:65536]                         # BGZF: a block inflates to at most 64 KiB
        for level in (0, 1, 4, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
                comp = raw_deflate(data, level, strategy)
                for at, g0 in ((0, 0), (18, 0), (1, 3), (2, 255), (3, 256), (7, 70001)):
                    rc, out = run(lib, comp, len(data), at, g0)
                    assert rc == 0, (len(data), level, strategy, at, g0, rc)
                    assert out == data
                    n += 1
    assert n > 3000


def test_small_windows_and_memlevels_many_blocks(lib):
    rng = np.random.default_rng(6)
    text = bytes(rng.choice(np.frombuffer(b"ACGTNacgtn\t\n0123456789", np.uint8), 65000))
    for wbits in (-9, -12, -15):
        for memlevel in (1, 3, 9):                  # memlevel 1: a new dynamic block every few hundred symbols
            comp = raw_deflate(text, 6, zlib.Z_DEFAULT_STRATEGY, wbits, memlevel)
            rc, out = run(lib, comp, len(text), 5, 77)
            assert rc == 0 and out == text


def test_wrong_size_and_truncation_fail_cleanly(lib):
    rng = np.random.default_rng(7)
    data = bytes(rng.integers(0, 7, 30000, dtype=np.uint8))
    comp = raw_deflate(data, 6)
    assert run(lib, comp, len(data) - 1)[0] != 0          # the trailer promised fewer bytes
    assert run(lib, comp, len(data) + 1)[0] != 0          # ... or more
    for cut in (1, 2, 10, len(comp) // 2, len(comp) - 1):
        rc, _ = run(lib, comp[:cut], len(data))
        assert rc != 0


def test_damaged_streams_never_disagree_with_zlib(lib):
    rng = np.random.default_rng(8)
    agree = fail = 0
    for trial in range(1500):
        data = bytes(rng.integers(0, int(rng.choice([2, 5, 64, 256])), int(rng.integers(1, 5000)), dtype=np.uint8))
        comp = bytearray(raw_deflate(data, int(rng.choice([1, 6, 9])), int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED]))))
        for _ in range(int(rng.integers(1, 4))):
            comp[int(rng.integers(0, len(comp)))] ^= 1 << int(rng.integers(0, 8))
        rc, out = run(lib, bytes(comp), len(data), int(rng.integers(0, 4)), int(rng.integers(0, 600)))
        if rc != 0:
            fail += 1
            continue
        # it decoded to exactly len(data) bytes: zlib must produce the same bytes from the same input
        d = zlib.decompressobj(-15)
        try:
            ref = d.decompress(bytes(comp))
        except zlib.error:
            ref = None
        assert ref is not None and ref[:len(data)] == out, trial
        agree += 1
    assert fail > 100 and agree > 10


def _fnv(b):
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.parametrize("form", INPUT_FORMS, ids=FORM_IDS)
def test_address_sanitizer_exact_buffers(form, tmp_path):
    """The same decoder under -fsanitize=address,undefined with EXACT-size buffers (input: the block + its 8-byte trailer + the
    16 bytes of padding the C ABI asks for; output: usize bytes). Covers what a fuzz without a sanitizer cannot see: reads
    beyond the padding. The crafted case is the one that used to run away: a dynamic block whose 1-bit code maps the zero
    bytes behind a truncated block to literals, usize = 65536 — it must stop with an error inside the padding."""
    exe = str(tmp_path / "inflate_asan")
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wno-unknown-pragmas"] + form + ["-o", exe,
                           os.path.join(ROOT, "tests", "inflate_asan_main.cpp"), os.path.join(ROOT, "tests", "inflate_host.cpp")])
    rng = np.random.default_rng(11)
    cases, expect = [], []
    # the runaway: 'A' and end-of-block as the only two literal/length codes (1 bit each), cut short
    aaaa = raw_deflate(b"A" * 65536, 6, zlib.Z_HUFFMAN_ONLY)
    for cut in (len(aaaa) // 2, len(aaaa) // 4, 40, len(aaaa) - 1):
        for at in (0, 1, 2, 3):
            cases.append((at, aaaa[:cut], 65536))
            expect.append(None)                              # must fail (and must not trip the sanitizer)
    cases.append((0, aaaa, 65536))
    expect.append(b"A" * 65536)
    # well-formed blocks of every kind, at every alignment
    for data in corpus(rng):
        data = data[:65536]
        for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_HUFFMAN_ONLY)):
            cases.append((int(rng.integers(0, 4)), raw_deflate(data, level, strategy), len(data)))
            expect.append(data)
    # damaged and truncated streams: any verdict, no bad access
    for trial in range(600):
        data = bytes(rng.integers(0, int(rng.choice([2, 5, 64, 256])), int(rng.integers(1, 5000)), dtype=np.uint8))
        comp = bytearray(raw_deflate(data, int(rng.choice([1, 6, 9])), int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY]))))
        if trial % 3 == 0:
            comp = comp[:int(rng.integers(1, len(comp) + 1))]
        else:
            for _ in range(int(rng.integers(1, 4))):
                comp[int(rng.integers(0, len(comp)))] ^= 1 << int(rng.integers(0, 8))
        usize = len(data) if trial % 5 else 65536
        cases.append((int(rng.integers(0, 4)), bytes(comp), usize))
        expect.append(False)                                 # False: don't care
    path = tmp_path / "cases.bin"
    with open(path, "wb") as f:
        f.write(np.uint32(len(cases)).tobytes())
        for at, comp, usize in cases:
            f.write(np.array([at, len(comp), usize], np.uint32).tobytes())
            f.write(comp)
    pr = subprocess.run([exe, str(path)], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = pr.stdout.split("\n")[:len(cases)]
    assert len(lines) == len(cases)
    for ln, want in zip(lines, expect):
        rc, h = ln.split()
        if want is None:
            assert int(rc) != 0
        elif want is not False:
            assert int(rc) == 0 and int(h) == _fnv(want)


class _Bits:
    """DEFLATE's bit order: fields LSB first, Huffman codes MSB first"""

    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def field(self, v, n):
        self.acc |= v << self.n
        self.n += n
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def code(self, c, n):
        self.field(int(format(c, f"0{n}b")[::-1], 2), n)

    def fixed_sym(self, s):                                      # RFC 1951 3.2.6
        if s < 144: self.code(0x30 + s, 8)
        elif s < 256: self.code(0x190 + s - 144, 9)
        elif s < 280: self.code(s - 256, 7)
        else: self.code(0xC0 + s - 280, 8)

    def done(self):
        if self.n:
            self.out.append(self.acc & 255)
        return bytes(self.out)


def test_code_lengths_staged_in_the_fullest_scratch_region(lib):
    """Pass 1 parks the code lengths of a block in the gap between the scratch region's literals and tokens while it builds the
    tables (itxi_tokens: `stage`). The gap is narrowest when a SECOND block's header is parsed behind a first block of nothing
    but three-byte matches (4 bytes of token per 3 of output), and the literal word in the making must survive it: hand-made
    streams, since no compressor writes them."""
    # (a) 'a', 21 800 matches of (3, distance 1), then a second fixed block of literals and a match reaching far back
    b = _Bits()
    b.field(0, 1); b.field(1, 2)
    b.fixed_sym(ord("a"))
    for _ in range(21800):
        b.fixed_sym(257)            # length 3
        b.code(0, 5)                # distance 1
    b.fixed_sym(256)
    b.field(1, 1); b.field(1, 2)
    for ch in b"bcd":
        b.fixed_sym(ch)
    b.fixed_sym(257); b.code(0, 5)  # ddd
    b.fixed_sym(256)
    comp = b.done()
    want = zlib.decompress(comp, -15)
    assert len(want) == 1 + 3 * 21800 + 3 + 3 and want.endswith(b"abcdddd")
    rc, got = run(lib, comp, len(want))
    assert rc == 0 and got == want
    # (b) 65 001 literals (one byte into a literal word), then a second block: the staged lengths start behind that word
    rng = np.random.default_rng(11)
    lits = bytes(rng.integers(0, 256, 65001, dtype=np.uint8))
    b = _Bits()
    b.field(0, 1); b.field(1, 2)
    for ch in lits:
        b.fixed_sym(ch)
    b.fixed_sym(256)
    b.field(1, 1); b.field(1, 2)
    for ch in b"xyz":
        b.fixed_sym(ch)
    b.fixed_sym(285); b.code(29, 5); b.field(8191, 13)          # length 258 from distance 32 768
    b.fixed_sym(256)
    comp = b.done()
    want = zlib.decompress(comp, -15)
    assert len(want) == 65001 + 3 + 258
    rc, got = run(lib, comp, len(want))
    assert rc == 0 and got == want
