// itx_inflate_core.h — raw DEFLATE (RFC 1951) of ONE BGZF block by ONE wavefront.
//
// What it replaces: the per-block zlib `inflate` of the reference's reader (cussamtools/bgzf.c:367-397 inflate_block,
// called from bgzf_read_block, bgzf.c:425-466) — a BAM is a sequence of independent gzip members of at most 64 KiB, so a
// chunk of the file is thousands of independent decodes.
//
// Shape on CDNA4: one wave per block, thousands of waves in flight. Huffman decoding is a serial bit chase, so the wave
// runs it as a UNIFORM program — all 64 lanes carry the same bit buffer and take the same branches (values the compiler
// keeps in scalar registers where it can) — and uses its width where the format is wide: match copies, stored blocks and
// the write-back move 64 bytes per step. Output is staged in an 8 KiB LDS ring addressed by the GLOBAL output offset, so
// every 256-byte stripe that fills up leaves as one coalesced 16-byte-per-lane store; matches read the ring (LDS is in
// order within a wave: no barriers), and the few that reach further back than the ring read what was already written
// back, behind a fence. Codes are decoded canonically from per-length counts held in registers (no table build beyond a
// counting sort of the symbols), after Mark Adler's puff.c description of the format.
//
// The same source compiles for the host with ITXI_WAVE == 1 (tests/inflate_host.cpp): the CPU test suite fuzzes it
// against zlib before any of it runs on a GPU. Every loop is bounded by input consumed or output produced; any
// inconsistency ends the block with a non-zero status and the caller decides (the host re-inflates such a block with
// zlib, whose verdict is the reference's).
#pragma once
#include <stdint.h>

#ifndef ITXI_WAVE
#error "define ITXI_WAVE (64 on the device, 1 on the host) and the ITXI_* hooks before including this file"
#endif

#define ITXI_RING 8192u                    // bytes of output kept in LDS; a power of two, a multiple of the stripe
#define ITXI_MASK (ITXI_RING - 1u)
#define ITXI_STRIPE 256u                   // write-back granule: 64 lanes x 4 bytes
#define ITXI_NEAR (ITXI_RING - 320u)       // matches up to this distance read the ring (a match writes at most 258 bytes ahead)

enum {
    ITXI_OK = 0,
    ITXI_E_INPUT = 1,       // ran past the block's compressed bytes
    ITXI_E_OUTPUT = 2,      // would write past the size the BGZF trailer promised, or ended short of it
    ITXI_E_BTYPE = 3,       // block type 3
    ITXI_E_STORED = 4,      // LEN != ~NLEN
    ITXI_E_CODES = 5,       // bad code-length set (counts, repeats, over-subscribed or incomplete)
    ITXI_E_SYMBOL = 6,      // a bit pattern that is no code, or length/distance symbol out of range
    ITXI_E_DIST = 7,        // distance reaches before the start of the block
};

struct ItxiLds {
    uint32_t ring32[ITXI_RING / 4];
    uint16_t lsym[288], dsym[32];          // symbols in canonical order (by code length, then symbol)
    uint16_t offs[16];
    uint8_t lens[352];                     // code lengths being set up: literal/length at 0, distance at 288, code-length code at 320
};

struct ItxiCodes {                         // counts per code length 1..15, two to a word (length 2k+1 low, 2k+2 high)
    uint32_t c[8];
};

struct ItxiIn {
    const uint32_t *w;                     // the compressed buffer as words (4-byte aligned base)
    uint32_t ip, end;                      // next word to load; byte offset of the first byte past the block's data
    uint64_t bb;                           // bit buffer, next bit in bit 0
    uint32_t bn;                           // valid bits in bb
};

ITXI_FN void itxi_in_start(ItxiIn &in, uint32_t byte_pos)
{
    in.ip = byte_pos >> 2;
    const uint32_t skip = (byte_pos & 3u) * 8u;
    in.bb = (uint64_t)(ITXI_LOADW(in.w, in.ip) >> skip);
    in.ip++;
    in.bn = 32u - skip;
}

// at least 33 valid bits afterwards; reading a few words past the end is harmless (the caller pads the buffer), consuming
// them is caught by the callers through itxi_overrun
ITXI_FN void itxi_refill(ItxiIn &in)
{
    if (in.bn <= 32u) {
        in.bb |= (uint64_t)ITXI_LOADW(in.w, in.ip) << in.bn;
        in.ip++;
        in.bn += 32u;
    }
}

ITXI_FN uint32_t itxi_bits(ItxiIn &in, uint32_t n)        // n <= 16, after a refill
{
    const uint32_t v = (uint32_t)in.bb & ((1u << n) - 1u);
    in.bb >>= n;
    in.bn -= n;
    return v;
}

ITXI_FN bool itxi_overrun(const ItxiIn &in)                 // a bit of a byte at or past `end` was consumed
{
    // bytes loaded so far end at 4 * ip; the last bn / 8 of them are untouched
    return in.ip * 4u - (in.bn >> 3) > in.end;
}

// One symbol of a canonical code (puff.c decode()): codes of each length are consecutive integers, shorter codes first.
// `peek` holds the next 15 stream bits; returns the symbol's index in canonical order and its length, or len = 0.
ITXI_FN uint32_t itxi_decode(const ItxiCodes &h, uint32_t peek, uint32_t &len_out)
{
    int32_t code = 0, first = 0, index = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int len = 2 * k + half + 1;
            if (len > 15) break;
            code |= (int32_t)(peek & 1u);
            peek >>= 1;
            const int32_t count = (int32_t)(half ? h.c[k] >> 16 : h.c[k] & 0xffffu);
            if (code - count < first) {
                len_out = (uint32_t)len;
                return (uint32_t)(index + (code - first));
            }
            index += count;
            first += count;
            first <<= 1;
            code <<= 1;
        }
    }
    len_out = 0;
    return 0;
}

// Counting sort of `n` code lengths into canonical order (puff.c construct()). Returns 0 for a complete code, > 0 for an
// incomplete one (bits left over), < 0 for an over-subscribed one. Every lane does the same thing to the same LDS words.
ITXI_FN int32_t itxi_construct(ItxiLds &S, ItxiCodes &h, uint16_t *sym, const uint8_t *lens, uint32_t n, uint32_t &n_zero_or_one)
{
    uint32_t cnt[16];
#pragma unroll
    for (int i = 0; i < 16; i++) cnt[i] = 0;
    for (uint32_t s = 0; s < n; s++) {
        const uint32_t l = ITXI_UNI((uint32_t)lens[s]);
#pragma unroll
        for (int i = 0; i < 16; i++) cnt[i] += (l == (uint32_t)i) ? 1u : 0u;
    }
    n_zero_or_one = cnt[0] + cnt[1];          // an incomplete code is legal only as ONE code of length 1 (zlib inftrees.c, puff.c)
    int32_t left = 1;
#pragma unroll
    for (int len = 1; len <= 15; len++) {
        left <<= 1;
        left -= (int32_t)cnt[len];
        if (left < 0) return left;                             // over-subscribed
    }
    uint32_t o = 0;
#pragma unroll
    for (int len = 1; len <= 15; len++) {
        S.offs[len] = (uint16_t)o;
        o += cnt[len];
    }
    for (uint32_t s = 0; s < n; s++) {
        const uint32_t l = ITXI_UNI((uint32_t)lens[s]);
        if (l != 0) {
            const uint32_t at = ITXI_UNI((uint32_t)S.offs[l]);
            sym[at] = (uint16_t)s;
            S.offs[l] = (uint16_t)(at + 1u);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) h.c[k] = cnt[2 * k + 1] | ((2 * k + 2 <= 15 ? cnt[2 * k + 2] : 0u) << 16);
    return left;
}

struct ItxiOut {
    uint8_t *g;                            // the whole output buffer
    uint32_t g0, gend;                     // this block's bytes are g[g0, gend)
    uint32_t gp;                           // next byte to produce
    uint32_t fl;                           // bytes below this offset are written back
};

// write back ring bytes [from, to): whole aligned stripes as words, ragged ends (a block's first and last bytes share
// words with its neighbours, which other waves write) byte by byte
ITXI_FN void itxi_writeback(ItxiLds &S, ItxiOut &o, uint32_t to, uint32_t lane)
{
    const uint8_t *ring8 = reinterpret_cast<const uint8_t *>(S.ring32);
    uint32_t from = o.fl;
    while (from < to) {
        const uint32_t stop = (from | (ITXI_STRIPE - 1u)) + 1u;                    // end of the stripe `from` lies in
        const uint32_t upto = stop < to ? stop : to;
        if ((from & (ITXI_STRIPE - 1u)) == 0 && upto == stop) {
            for (uint32_t k = lane; k < ITXI_STRIPE / 4; k += ITXI_WAVE)
                reinterpret_cast<uint32_t *>(o.g)[(from >> 2) + k] = S.ring32[((from & ITXI_MASK) >> 2) + k];
        } else {
            for (uint32_t a = from + lane; a < upto; a += ITXI_WAVE) o.g[a] = ring8[a & ITXI_MASK];
        }
        from = upto;
    }
    o.fl = to;
}

// after gp moved: every stripe that is complete leaves
ITXI_FN void itxi_flush_full(ItxiLds &S, ItxiOut &o, uint32_t lane)
{
    const uint32_t full = o.gp & ~(ITXI_STRIPE - 1u);
    if (full > o.fl) itxi_writeback(S, o, full, lane);
}

ITXI_FN int itxi_block(ItxiLds &S, const uint32_t *comp_words, uint32_t data_pos, uint32_t data_end, uint8_t *out, uint32_t g0, uint32_t usize,
                       uint32_t lane)
{
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const uint8_t clorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

    uint8_t *ring8 = reinterpret_cast<uint8_t *>(S.ring32);
    ItxiIn in;
    in.w = comp_words;
    in.end = data_end;
    itxi_in_start(in, data_pos);
    ItxiOut o;
    o.g = out;
    o.g0 = g0;
    o.gend = g0 + usize;
    o.gp = g0;
    o.fl = g0;
    ItxiCodes lc, dc;

    for (;;) {                                                     // one DEFLATE block per turn; each consumes >= 3 bits
        itxi_refill(in);
        const uint32_t last = itxi_bits(in, 1);
        const uint32_t type = itxi_bits(in, 2);
        if (itxi_overrun(in)) return ITXI_E_INPUT;
        if (type == 0) {
            // stored: skip to the byte boundary, LEN, NLEN, LEN bytes
            itxi_bits(in, in.bn & 7u);
            itxi_refill(in);
            const uint32_t len = itxi_bits(in, 16);
            itxi_refill(in);
            const uint32_t nlen = itxi_bits(in, 16);
            if ((len ^ nlen) != 0xffffu) return ITXI_E_STORED;
            const uint32_t pos = in.ip * 4u - in.bn / 8u;          // byte position of the data (bn is a multiple of 8 here)
            if (pos + len > data_end) return ITXI_E_INPUT;
            if (len > o.gend - o.gp) return ITXI_E_OUTPUT;
            const uint8_t *src = reinterpret_cast<const uint8_t *>(comp_words) + pos;
            uint32_t done = 0;
            while (done < len) {                                   // a stripe's worth at a time so the ring never laps itself
                uint32_t n = len - done;
                const uint32_t room = ITXI_STRIPE - (o.gp & (ITXI_STRIPE - 1u));
                if (n > room) n = room;
                for (uint32_t k = lane; k < n; k += ITXI_WAVE) ring8[(o.gp + k) & ITXI_MASK] = ITXI_LOADB(src, done + k);
                o.gp += n;
                done += n;
                itxi_flush_full(S, o, lane);
            }
            itxi_in_start(in, pos + len);
        } else if (type == 3) {
            return ITXI_E_BTYPE;
        } else {
            uint32_t nl, nd;
            if (type == 1) {
                nl = 288;
                nd = 30;
                for (uint32_t s = lane; s < 288; s += ITXI_WAVE) S.lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
                for (uint32_t s = lane; s < 30; s += ITXI_WAVE) S.lens[288 + s] = 5;
            } else {
                itxi_refill(in);
                nl = itxi_bits(in, 5) + 257u;
                nd = itxi_bits(in, 5) + 1u;
                const uint32_t nc = itxi_bits(in, 4) + 4u;
                if (nl > 286u || nd > 30u) return ITXI_E_CODES;
                for (uint32_t s = lane; s < 19; s += ITXI_WAVE) S.lens[320 + s] = 0;
                for (uint32_t i = 0; i < nc; i++) {
                    itxi_refill(in);
                    S.lens[320 + clorder[i]] = (uint8_t)itxi_bits(in, 3);
                }
                if (itxi_overrun(in)) return ITXI_E_INPUT;
                ItxiCodes cc;
                uint32_t nz;
                if (itxi_construct(S, cc, S.lsym, S.lens + 320, 19, nz) != 0) return ITXI_E_CODES;     // must be complete
                // code lengths of the literal/length and distance codes, run-length coded as ONE sequence (a run may
                // cross from one code into the other); entry idx of it lives at place(idx)
#define ITXI_PLACE(i) ((i) < nl ? (i) : 288u + ((i) - nl))
                uint32_t idx = 0;
                while (idx < nl + nd) {
                    itxi_refill(in);
                    uint32_t cl;
                    const uint32_t at = itxi_decode(cc, (uint32_t)in.bb & 0x7fffu, cl);
                    if (cl == 0) return ITXI_E_SYMBOL;
                    itxi_bits(in, cl);
                    const uint32_t sym = ITXI_UNI((uint32_t)S.lsym[at]);
                    if (sym < 16) {
                        S.lens[ITXI_PLACE(idx)] = (uint8_t)sym;
                        idx++;
                    } else {
                        uint32_t rep, val = 0;
                        if (sym == 16) {
                            if (idx == 0) return ITXI_E_CODES;
                            val = ITXI_UNI((uint32_t)S.lens[ITXI_PLACE(idx - 1)]);
                            rep = 3u + itxi_bits(in, 2);
                        } else if (sym == 17) {
                            rep = 3u + itxi_bits(in, 3);
                        } else {
                            rep = 11u + itxi_bits(in, 7);
                        }
                        if (idx + rep > nl + nd) return ITXI_E_CODES;
                        for (uint32_t k = lane; k < rep; k += ITXI_WAVE) S.lens[ITXI_PLACE(idx + k)] = (uint8_t)val;
                        idx += rep;
                    }
                    if (itxi_overrun(in)) return ITXI_E_INPUT;
                }
#undef ITXI_PLACE
                if (ITXI_UNI((uint32_t)S.lens[256]) == 0) return ITXI_E_CODES;                   // no end-of-block code
            }
            uint32_t nz;
            int32_t left = itxi_construct(S, lc, S.lsym, S.lens, nl, nz);
            if (type == 2 && left != 0 && (left < 0 || nl != nz)) return ITXI_E_CODES;           // the fixed code is incomplete by definition
            left = itxi_construct(S, dc, S.dsym, S.lens + 288, nd, nz);
            if (type == 2 && left != 0 && (left < 0 || nd != nz)) return ITXI_E_CODES;

            for (;;) {                                             // one symbol per turn; each produces output or ends the block
                itxi_refill(in);
                uint32_t cl;
                uint32_t at = itxi_decode(lc, (uint32_t)in.bb & 0x7fffu, cl);
                if (cl == 0) return ITXI_E_SYMBOL;
                itxi_bits(in, cl);
                const uint32_t sym = ITXI_UNI((uint32_t)S.lsym[at]);
                if (sym < 256u) {
                    if (o.gp >= o.gend) return ITXI_E_OUTPUT;
                    if (lane == 0) ring8[o.gp & ITXI_MASK] = (uint8_t)sym;
                    o.gp++;
                    if ((o.gp & (ITXI_STRIPE - 1u)) == 0) {
                        if (itxi_overrun(in)) return ITXI_E_INPUT;
                        itxi_flush_full(S, o, lane);
                    }
                    continue;
                }
                if (sym == 256u) break;
                const uint32_t li = sym - 257u;
                if (li >= 29u) return ITXI_E_SYMBOL;
                const uint32_t len = lbase[li] + itxi_bits(in, lext[li]);
                itxi_refill(in);
                at = itxi_decode(dc, (uint32_t)in.bb & 0x7fffu, cl);
                if (cl == 0) return ITXI_E_SYMBOL;
                itxi_bits(in, cl);
                const uint32_t di = ITXI_UNI((uint32_t)S.dsym[at]);
                if (di >= 30u) return ITXI_E_SYMBOL;
                const uint32_t dist = dbase[di] + itxi_bits(in, dext[di]);
                if (itxi_overrun(in)) return ITXI_E_INPUT;
                if (dist > o.gp - o.g0) return ITXI_E_DIST;
                if (len > o.gend - o.gp) return ITXI_E_OUTPUT;
                const uint32_t src0 = o.gp - dist;
                if (dist <= ITXI_NEAR) {
                    if (dist >= 64u || dist >= len) {             // 64: the widest step any build takes
                        // a step's sources were all written before the step (earlier steps or earlier symbols)
                        for (uint32_t k0 = 0; k0 < len; k0 += ITXI_WAVE) {
                            const uint32_t k = k0 + lane;
                            if (k < len) ring8[(o.gp + k) & ITXI_MASK] = ring8[(src0 + k) & ITXI_MASK];
                        }
                    } else {
                        // the match overlaps itself within a step: the output is the last `dist` bytes repeated
                        for (uint32_t k0 = 0; k0 < len; k0 += ITXI_WAVE) {
                            const uint32_t k = k0 + lane;
                            if (k < len) ring8[(o.gp + k) & ITXI_MASK] = ring8[(src0 + k % dist) & ITXI_MASK];
                        }
                    }
                } else {
                    // further back than the ring reaches: those bytes left in whole stripes long ago (dist > 7 KiB > len)
                    ITXI_FENCE();
                    for (uint32_t k0 = 0; k0 < len; k0 += ITXI_WAVE) {
                        const uint32_t k = k0 + lane;
                        if (k < len) ring8[(o.gp + k) & ITXI_MASK] = ITXI_LOADB(o.g, src0 + k);
                    }
                }
                o.gp += len;
                itxi_flush_full(S, o, lane);
            }
        }
        if (last) break;
        if (itxi_overrun(in)) return ITXI_E_INPUT;
    }
    if (itxi_overrun(in)) return ITXI_E_INPUT;
    if (o.gp != o.gend) return ITXI_E_OUTPUT;
    itxi_writeback(S, o, o.gend, lane);
    return ITXI_OK;
}
