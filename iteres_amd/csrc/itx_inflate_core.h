// itx_inflate_core.h — raw DEFLATE (RFC 1951) of BGZF blocks on a GPU, in two passes.
//
// What it replaces: the per-block zlib `inflate` of the reference's reader (cussamtools/bgzf.c:367-397 inflate_block,
// called from bgzf_read_block, bgzf.c:425-466) — a BAM is a sequence of independent gzip members of at most 64 KiB, so a
// chunk of the file is thousands of independent decodes.
//
// DEFLATE has two halves of very different shape, and each gets the mapping that suits it:
//
//   pass 1, itxi_tokens — the Huffman half is a serial bit chase per block with no use for a wide machine inside one
//     block (a wave running it as one uniform program is bound by the CU's one-scalar-instruction-per-cycle issue:
//     measured 8 GB/s on the chip). So here every LANE decodes its own block: 64 blocks per wave, all state in the lane's
//     registers (bit buffer, canonical-code counts), the symbol tables lane-interleaved in LDS (bank = lane: no
//     conflicts whatever the symbols). It needs no output window: it writes the block's LITERALS as a byte string and
//     its MATCHES as (literals before, length, distance) tokens.
//   pass 2, itxi_resolve — the LZ77 half is memory movement: one WAVE per block replays the tokens, 64 bytes per step,
//     in an 8 KiB LDS ring addressed by the global output offset; every 256-byte stripe that fills up leaves as one
//     coalesced store. Matches read the ring (LDS is in order within a wave: no barriers); the few that reach further
//     back read what was already written back, behind a fence.
//
// Codes are decoded canonically from per-length counts (a counting sort of the symbols is the whole table build), after
// Mark Adler's puff.c description of the format.
//
// The same source compiles for the host (tests/inflate_host.cpp: one-lane "wave", plain arrays): the CPU suite fuzzes it
// against zlib before any of it runs on a GPU. Every loop is bounded by input consumed or output produced; any
// inconsistency ends the block with a non-zero status and the caller decides (the host re-inflates such a block with
// zlib, whose verdict is the reference's).
//
// Hooks the including file defines: ITXI_FN (function attributes), ITXI_WAVE (lanes that share pass 2: 64 / 1),
// ITXI_UNI(x) (pass 2: a wave-uniform value as such), ITXI_AT(p, i) (pass 1: element i of a per-decoder table: lane-
// interleaved on the device, using the lane id `ln` in scope), ITXI_BCAST(v, j) (pass 2: lane j's v, for all; j uniform),
// ITXI_LANE_READ(v, j) (pass 2: lane j's v with j per lane), ITXI_BALLOT(p) (pass 2: 64-bit mask of the lanes where p holds),
// ITXI_MBCNT(m, lane) (set bits of m below `lane`), ITXI_LDS_OR(ptr, v) (pass 2: atomic OR into the wave's LDS),
// ITXI_BITREV32(x), ITXI_PKSIGN16(a, b) (bit 15 of a - b in each 16-bit half, moved to bits 0 and 16), ITXI_LOADW / ITXI_LOADB
// (global loads), ITXI_FENCE(), ITXI_SCAN_ADD(v, lane) (pass 2: inclusive prefix sum of v over the lanes).
#pragma once
#include <stdint.h>

#ifndef ITXI_WAVE
#error "define the ITXI_* hooks before including this file"
#endif

// timing-only experiment builds (tools/build_variant.sh): -DITXI_EXP_NOSTORE leaves pass 1's token and literal stores out
#ifdef ITXI_EXP_NOSTORE
#define ITXI_EXP_STORE(...) ((void)0)
#else
#define ITXI_EXP_STORE(...) __VA_ARGS__
#endif

#ifndef ITXI_RING
#define ITXI_RING 4096u                    // pass 2: bytes of output kept in LDS; a power of two, a multiple of the stripe. 6 KB of LDS per
                                           // wave with the literal stage: 26 waves per CU instead of 13 with 8 + 4 KB — pass 2 is a chain of LDS
                                           // round trips per token, so resident waves are its throughput (measured: 8.5 ms per 8 k blocks, was 13)
#endif
#define ITXI_MASK (ITXI_RING - 1u)
#define ITXI_STRIPE 256u                   // write-back granule: 64 lanes x 4 bytes
#define ITXI_NEAR (ITXI_RING - 320u)       // matches up to this distance read the ring (a match writes at most 258 bytes ahead)
#define ITXI_LAG 1024u                     // pass 2: full stripes may wait this long for their write-back (+ a stripe + a token < ITXI_NEAR - 258)
#ifndef ITXI_LITS
#define ITXI_LITS 2                        // pass 1: literal/length codes a turn may take (itxi_tokens)
#endif
#define ITXI_MAX_BLOCK 65536u              // BGZF: a block inflates to at most 64 KiB
// What pass 1 leaves for pass 2 lives in ONE region of scratch per block: the literal bytes grow up from its bottom, the 4-byte
// tokens down from its top. A match stands for at least 3 bytes of output and takes 4 bytes of token, a literal takes one, a
// run token (below) is written once per 256 literals at most: literals + 4 x tokens <= 65536 x 4 / 3 at any moment, so the
// two never meet in 88 KiB (a quarter megabyte per block when the two had arrays of their own, sized for their worst cases).
#define ITXI_REGION 90112u                 // bytes per block, a multiple of 16
#define ITXI_MAX_TOK (ITXI_MAX_BLOCK / 3u + 1u + ITXI_MAX_BLOCK / 256u + 2u)
// a token: a match as (length - 3) | (distance - 1) << 8 | (literals since the previous token, at most 255) << 23; a longer run of
// literals gets a token of its own in front: bit 31 | the run
#define ITXI_TOK_RUN 0x80000000u
#define ITXI_BMAP 2048u                    // pass 2: bytes of output the fast steps of one batch of tokens may cover (a bitmap of token starts in LDS)

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

// ---------------------------------------------------------------------------------------------------- pass 1: tokens

// One decoder's tables (element i through ITXI_AT), 420 bytes: on the device they live in LDS, a lane's worth per block, and how
// many blocks pass 1 can hold at once is how much of it fits — 26.25 KB per wave of 64 blocks, SIX waves per CU (with the code
// lengths and the counting sort's cursors in arrays of their own it was 40 KB and four; with 16-bit symbols and a byte per code
// length 68 KB and two). The code lengths being set up (two to a byte: literal/length at 0, distance at 288, code-length code
// at 320) borrow the symbol table's room from byte ITXI_LENS_AT on — the header is parsed before there are symbols to keep, the
// code-length code's own 19 symbols sit below — and move to a few words of the block's scratch region before the tables are
// built from them (itxi_tokens: `stage`).
#define ITXI_LENS_AT 32u
#ifndef ITXI_SYM16
struct ItxiTab {
    uint8_t *lsym8;                        // [288]: literal/length symbols in canonical order (by code length, then symbol), low 8 bits
    uint32_t *lhi;                         // [9]:   bit i of the bitmap: the symbol at canonical place i is >= 256
    uint8_t *dsym;                         // [32]:  distance symbols in canonical order
    uint16_t *loffs, *doffs;               // [16] each: itxi_decode's per-length offsets of the two codes in use (the counting sort's cursors meanwhile)
};
#define ITXI_LEN_GET(T, i) ((uint32_t)(ITXI_AT((T).lsym8, ITXI_LENS_AT + ((i) >> 1)) >> (((i) & 1u) * 4u)) & 15u)
#define ITXI_LEN_SET(T, i, v)                                                                                     \
    do {                                                                                                          \
        const uint32_t i__ = (i), sh__ = (i__ & 1u) * 4u, at__ = ITXI_LENS_AT + (i__ >> 1);                       \
        ITXI_AT((T).lsym8, at__) = (uint8_t)((ITXI_AT((T).lsym8, at__) & ~(15u << sh__)) | (((v) & 15u) << sh__)); \
    } while (0)
#define ITXI_LSYM_PUT(T, at, sy)                                                    \
    do {                                                                            \
        ITXI_AT((T).lsym8, (at)) = (uint8_t)(sy);                                   \
        if ((sy) >= 256u) ITXI_AT((T).lhi, (at) >> 5) |= 1u << ((at) & 31u);        \
    } while (0)
#define ITXI_LSYM_GET(T, at) ((uint32_t)ITXI_AT((T).lsym8, (at)) | (((ITXI_AT((T).lhi, (at) >> 5) >> ((at) & 31u)) & 1u) << 8))
#define ITXI_LSYM_GET_LOW(T, at) ((uint32_t)ITXI_AT((T).lsym8, (at)))      // a symbol known to be below 256 (the code-length code's)
#define ITXI_LSYM_CLEAR(T)                                            \
    do {                                                              \
        for (uint32_t k__ = 0; k__ < 9; k__++) ITXI_AT((T).lhi, k__) = 0; \
    } while (0)
#else
// Experiment builds (-DITXI_SYM16): the literal/length symbols as ONE 16-bit table — a symbol costs one LDS read instead of two
// (its low byte and its ninth bit), the tables 43 KB per wave instead of 26 (three waves to a CU). The code lengths being set up
// lie four to an element from element ITXI_LENS_AT on.
struct ItxiTab {
    uint16_t *lsym16;                      // [288]: literal/length symbols in canonical order
    uint8_t *dsym;
    uint16_t *loffs, *doffs;
};
#define ITXI_LEN_GET(T, i) ((uint32_t)(ITXI_AT((T).lsym16, ITXI_LENS_AT + ((i) >> 2)) >> (((i) & 3u) * 4u)) & 15u)
#define ITXI_LEN_SET(T, i, v)                                                                                         \
    do {                                                                                                              \
        const uint32_t i__ = (i), sh__ = (i__ & 3u) * 4u, at__ = ITXI_LENS_AT + (i__ >> 2);                           \
        ITXI_AT((T).lsym16, at__) = (uint16_t)((ITXI_AT((T).lsym16, at__) & ~(15u << sh__)) | (((v) & 15u) << sh__)); \
    } while (0)
#define ITXI_LSYM_PUT(T, at, sy) (ITXI_AT((T).lsym16, (at)) = (uint16_t)(sy))
#define ITXI_LSYM_GET(T, at) ((uint32_t)ITXI_AT((T).lsym16, (at)))
#define ITXI_LSYM_GET_LOW(T, at) ((uint32_t)ITXI_AT((T).lsym16, (at)))
#define ITXI_LSYM_CLEAR(T) ((void)0)
#endif
#define ITXI_STAGE_WORDS 40u               // the code lengths as words, eight to a word (literal/length from word 0, distance from word 36)

struct ItxiCodes {
    // bound l (l = 1..15): every code of length <= l, written MSB first and left-aligned to 15 bits, is below it (the bounds
    // rise with l, up to 2^15; bound 15 < 2^15 exactly when the code is incomplete). Bounds 1..14 sit two to a word, as
    // 16-bit halves, for packed 16-bit arithmetic; bound 15 by itself.
    uint32_t lp[7];
    uint32_t lim15;
};

struct ItxiIn {
    const uint32_t *w;                     // the compressed buffer as words (16-byte aligned base)
    uint32_t ip, end;                      // next word to enter the bit buffer; byte offset of the first byte past the block's data
    uint32_t lim;                          // end / 4 + 2: a well-formed block never takes a word beyond this one (see itxi_past)
    uint64_t bb;                           // bit buffer, next bit in bit 0
    uint32_t bn;                           // valid bits in bb
    // Read-ahead: the words ip, ip + 1, .. (na of them, a0 first) are in registers, and the four after those (b0..b3, word
    // index nb) are on their way. A lane that decodes alone on its SIMD cannot hide a load behind other waves: with one word
    // of look-ahead every match — which takes up to 48 bits, two refills — waited for memory once. With 16 to 32 bytes of
    // look-ahead (a dozen symbols) the next batch has long arrived when the current one is used up.
    uint32_t a0, a1, a2, a3, na;
    uint32_t b0, b1, b2, b3, nb;
    uint32_t stop;                         // batches at or beyond this word index lie past the block's last 16-byte line: they read as zeros
};

#ifndef ITXI_LOAD4
#define ITXI_LOAD4(w, i, a, b, c, d)   \
    do {                               \
        (a) = ITXI_LOADW(w, (i));      \
        (b) = ITXI_LOADW(w, (i) + 1u); \
        (c) = ITXI_LOADW(w, (i) + 2u); \
        (d) = ITXI_LOADW(w, (i) + 3u); \
    } while (0)
#endif

// the batch of four words at word index i (a multiple of four) into b0..b3; nothing past the block's last 16-byte line is
// touched: at most 15 bytes beyond `end` are ever read (a BGZF block's 8-byte trailer and the caller's 16 bytes of padding
// follow it), and what lies further reads as zeros — consuming those is caught by itxi_overrun / itxi_past
ITXI_FN void itxi_batch(ItxiIn &in, uint32_t i)
{
    in.nb = i;
    in.b0 = in.b1 = in.b2 = in.b3 = 0;
    if (i < in.stop) ITXI_LOAD4(in.w, i, in.b0, in.b1, in.b2, in.b3);
}

#ifdef ITXI_SIMPLE_IN
// One word of look-ahead instead of the FIFO: the word at ip sits in a0, its successor is asked for when a0 is taken. Fewer
// registers to carry through the symbol loop (the FIFO's eight words are shuffled on every take), one 4-byte load per refill.
// The load is unconditional (its index clamped to the block's last word) and its value is masked only when it is TAKEN: a
// load under a condition is waited for where it is issued (the compiler merges it with the zero of the other branch), and
// the look-ahead would hide nothing.
ITXI_FN void itxi_ahead(ItxiIn &in)
{
    const bool inside = in.ip < in.stop;
    in.a0 = ITXI_LOADW(in.w, inside ? in.ip : in.stop - 1u);
    in.a1 = inside ? 0xffffffffu : 0u;
}
ITXI_FN uint32_t itxi_take(ItxiIn &in)
{
    const uint32_t w = in.a0 & in.a1;
    in.ip++;
    itxi_ahead(in);
    return w;
}
ITXI_FN void itxi_in_start(ItxiIn &in, uint32_t byte_pos)
{
    in.stop = (in.end + 3u) >> 2;                                  // words that hold bytes of the block (>= 1); what lies behind reads as zeros
    in.ip = byte_pos >> 2;
    itxi_ahead(in);
    const uint32_t skip = (byte_pos & 3u) * 8u;
    in.bb = (uint64_t)(itxi_take(in) >> skip);
    in.bn = 32u - skip;
}
#else
ITXI_FN uint32_t itxi_take(ItxiIn &in)                      // the word at ip; keeps the read-ahead going
{
    const uint32_t w = in.a0;
    in.a0 = in.a1;
    in.a1 = in.a2;
    in.a2 = in.a3;
    in.na--;
    if (in.na == 0) {
        in.a0 = in.b0;
        in.a1 = in.b1;
        in.a2 = in.b2;
        in.a3 = in.b3;
        in.na = 4;
        itxi_batch(in, in.nb + 4u);
    }
    in.ip++;
    return w;
}

ITXI_FN void itxi_in_start(ItxiIn &in, uint32_t byte_pos)
{
    in.stop = ((in.end + 15u) & ~15u) >> 2;
    const uint32_t first = byte_pos >> 2, base = first & ~3u;
    itxi_batch(in, base);
    in.a0 = in.b0;
    in.a1 = in.b1;
    in.a2 = in.b2;
    in.a3 = in.b3;
    in.na = 4;
    itxi_batch(in, base + 4u);
    in.ip = base;
    for (uint32_t k = base; k < first; k++) (void)itxi_take(in);          // the words of the line in front of the data
    const uint32_t skip = (byte_pos & 3u) * 8u;
    in.bb = (uint64_t)(itxi_take(in) >> skip);
    in.bn = 32u - skip;
}
#endif

// at least 33 valid bits afterwards
ITXI_FN void itxi_refill(ItxiIn &in)
{
    if (in.bn <= 32u) {
        in.bb |= (uint64_t)itxi_take(in) << in.bn;
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

// The cheap bound for loops that consume input without producing a reason to stop (a run of literals): more words have
// entered the bit buffer than a well-formed block can have taken. A valid block has 4 * ip - bn / 8 <= end with bn <= 64
// wherever it is tested, so ip <= end / 4 + 2 = lim there. (What is READ is bounded separately, by itxi_batch.)
ITXI_FN bool itxi_past(const ItxiIn &in) { return in.ip > in.lim; }

// One symbol of a canonical code: codes of each length are consecutive integers, shorter codes first (RFC 1951 3.2.2), so
// with the next 15 stream bits read MSB first as the number v, the code's length is 1 + the number of bounds v has
// reached — fifteen independent compares instead of puff.c's bit-by-bit walk (one wave per SIMD: a dependent chain
// costs its full latency) — and its place in canonical order is v's leading `len` bits plus a per-length offset
// (`offs`, filled by itxi_construct). Returns that place and the length, or len = 0 for a pattern that is no code.
ITXI_FN uint32_t itxi_decode(const ItxiCodes &h, const uint16_t *offs, uint32_t ln, uint32_t peek, uint32_t &len_out)
{
    (void)ln;
    const uint32_t v = ITXI_BITREV32(peek) >> 17;                  // 15 bits, first stream bit on top
    const uint32_t vv = v | v << 16;
    // v - bound in 16-bit halves: both are at most 2^15, so the difference's bit 15 says v < bound (also for bound = 2^15)
    uint32_t below = 0;                                            // per half: how many bounds v has NOT reached
#pragma unroll
    for (int k = 0; k < 7; k++) below += ITXI_PKSIGN16(vv, h.lp[k]);
    const uint32_t len = 15u - ((below & 0xffffu) + (below >> 16));       // 1 + the number of bounds 1..14 reached
    const bool ok = v < h.lim15;
    const uint32_t at = (uint32_t)(uint16_t)(ITXI_AT(offs, len) + (v >> (15u - len)));
    len_out = ok ? len : 0u;
    return ok ? at : 0u;                                           // a pattern that is no code must not become a table index
}

// eight code lengths (nibbles 8 * widx ..) as one word: from the staged copy, or from where the header parse keeps them
ITXI_FN uint32_t itxi_lens_word(const ItxiTab &T, uint32_t ln, const uint32_t *stage, uint32_t widx)
{
    (void)ln;
    if (stage) return stage[widx];
#ifndef ITXI_SYM16
    const uint32_t b = ITXI_LENS_AT + 4u * widx;
    return (uint32_t)ITXI_AT(T.lsym8, b) | (uint32_t)ITXI_AT(T.lsym8, b + 1u) << 8 | (uint32_t)ITXI_AT(T.lsym8, b + 2u) << 16 | (uint32_t)ITXI_AT(T.lsym8, b + 3u) << 24;
#else
    const uint32_t b = ITXI_LENS_AT + 2u * widx;
    return (uint32_t)ITXI_AT(T.lsym16, b) | (uint32_t)ITXI_AT(T.lsym16, b + 1u) << 16;
#endif
}

// Counting sort of `n` code lengths (nibbles base .., base a multiple of 8; `stage`: read from there, NULL: from the tables'
// own room) into canonical order (puff.c construct()). Returns 0 for a complete code, > 0 for an incomplete one (bits left
// over), < 0 for an over-subscribed one.
ITXI_FN int32_t itxi_construct(const ItxiTab &T, uint32_t ln, ItxiCodes &h, bool dist, uint16_t *offs_out, const uint32_t *stage, uint32_t base, uint32_t n,
                               uint32_t &n_zero_or_one)
{
    (void)ln;
    uint32_t cnt[16];
#pragma unroll
    for (int i = 0; i < 16; i++) cnt[i] = 0;
    const uint32_t w0 = base >> 3;
    for (uint32_t w = 0; 8u * w < n; w++) {
        const uint32_t word = itxi_lens_word(T, ln, stage, w0 + w);
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t l = 8u * w + k < n ? (word >> (4u * k)) & 15u : 16u;
#pragma unroll
            for (int i = 0; i < 16; i++) cnt[i] += (l == (uint32_t)i) ? 1u : 0u;
        }
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
        ITXI_AT(offs_out, len) = (uint16_t)o;                  // the sort's cursors; the decoder's offsets replace them below
        o += cnt[len];
    }
    if (!dist) ITXI_LSYM_CLEAR(T);
    for (uint32_t w = 0; 8u * w < n; w++) {
        const uint32_t word = itxi_lens_word(T, ln, stage, w0 + w);
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t sy = 8u * w + k;
            const uint32_t l = sy < n ? (word >> (4u * k)) & 15u : 0u;
            if (l != 0) {
                const uint32_t at = ITXI_AT(offs_out, l);
                if (dist) {
                    ITXI_AT(T.dsym, at) = (uint8_t)sy;
                } else {
                    ITXI_LSYM_PUT(T, at, sy);
                }
                ITXI_AT(offs_out, l) = (uint16_t)(at + 1u);
            }
        }
    }
    // what itxi_decode wants: the bounds, and per length (place of its first code) - (its first code), mod 2^16
    uint32_t first = 0, index = 0;
#pragma unroll
    for (int len = 1; len <= 15; len++) {
        ITXI_AT(offs_out, len) = (uint16_t)(index - first);
        index += cnt[len];
        first += cnt[len];
        const uint32_t bound = first << (15 - len);
        if (len == 15) h.lim15 = bound;
        else if ((len - 1) & 1) h.lp[(len - 1) >> 1] |= bound << 16;
        else h.lp[(len - 1) >> 1] = bound;
        first <<= 1;
    }
    return left;
}

// What pass 1 leaves for pass 2, per block (one region, see ITXI_REGION): lit[0, n_lit) the literal bytes in stream order;
// tok_top[-1 - i], i in [0, n_tok): the tokens; literals after the last token: n_lit - (sum of the tokens' runs). A stored
// block's bytes are literals.
struct ItxiTokens {
    uint8_t *lit;                          // bottom of the region (16-byte aligned)
    uint32_t *tok_top;                     // its top
    uint32_t n_lit, n_tok;
};

ITXI_FN int itxi_tokens(const ItxiTab &T, uint32_t ln, const uint32_t *comp_words, uint32_t data_pos, uint32_t data_end, uint32_t usize, ItxiTokens &K)
{
    static const uint8_t clorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

    ItxiIn in;
    in.w = comp_words;
    in.end = data_end;
    in.lim = data_end / 4u + 2u;
    itxi_in_start(in, data_pos);
    uint32_t produced = 0;                 // bytes the tokens so far stand for
    uint32_t n_lit = 0, n_tok = 0, run = 0;       // run: literals since the last match
    // literals leave four at a time (a lane's stores are to its own block's scratch: every store instruction of the wave
    // touches as many cache lines as it has active lanes, so fewer, wider stores are what counts), matches as one 8-byte pair
    uint32_t lacc = 0;
    uint32_t *lit32 = reinterpret_cast<uint32_t *>(K.lit);
#define ITXI_PUT_LIT(byte)                                   \
    do {                                                     \
        lacc |= (uint32_t)(byte) << ((n_lit & 3u) * 8u);     \
        n_lit++;                                             \
        if ((n_lit & 3u) == 0) {                             \
            ITXI_EXP_STORE(lit32[(n_lit >> 2) - 1u] = lacc); \
            lacc = 0;                                        \
        }                                                    \
    } while (0)
    K.n_lit = K.n_tok = 0;
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
            if (len > usize - produced) return ITXI_E_OUTPUT;
            const uint8_t *src = reinterpret_cast<const uint8_t *>(comp_words) + pos;
            for (uint32_t k = 0; k < len; k++) ITXI_PUT_LIT(ITXI_LOADB(src, k));
            run += len;
            produced += len;
            itxi_in_start(in, pos + len);
        } else if (type == 3) {
            return ITXI_E_BTYPE;
        } else {
            uint32_t nl, nd;
            if (type == 1) {
                nl = 288;
                nd = 30;
                for (uint32_t s = 0; s < 288; s++) ITXI_LEN_SET(T, s, (s < 144 ? 8u : s < 256 ? 9u : s < 280 ? 7u : 8u));
                for (uint32_t s = 0; s < 30; s++) ITXI_LEN_SET(T, 288 + s, 5u);
            } else {
                itxi_refill(in);
                nl = itxi_bits(in, 5) + 257u;
                nd = itxi_bits(in, 5) + 1u;
                const uint32_t nc = itxi_bits(in, 4) + 4u;
                if (nl > 286u || nd > 30u) return ITXI_E_CODES;
                for (uint32_t s = 0; s < 19; s++) ITXI_LEN_SET(T, 320 + s, 0u);
                for (uint32_t i = 0; i < nc; i++) {
                    itxi_refill(in);
                    ITXI_LEN_SET(T, 320 + clorder[i], itxi_bits(in, 3));
                }
                if (itxi_overrun(in)) return ITXI_E_INPUT;
                ItxiCodes cc;
                uint32_t n01;
                if (itxi_construct(T, ln, cc, false, T.loffs, nullptr, 320, 19, n01) != 0) return ITXI_E_CODES; // must be complete (its 19 symbols go to lsym8[0, 19): below the lengths)
                // code lengths of the literal/length and distance codes, run-length coded as ONE sequence (a run may
                // cross from one code into the other); entry idx of it lives at place(idx)
#define ITXI_PLACE(i) ((i) < nl ? (i) : 288u + ((i) - nl))
                uint32_t idx = 0;
                while (idx < nl + nd) {
                    itxi_refill(in);
                    uint32_t cl;
                    const uint32_t at = itxi_decode(cc, T.loffs, ln, (uint32_t)in.bb & 0x7fffu, cl);
                    if (cl == 0) return ITXI_E_SYMBOL;
                    itxi_bits(in, cl);
                    const uint32_t sym = ITXI_LSYM_GET_LOW(T, at);
                    if (sym < 16) {
                        ITXI_LEN_SET(T, ITXI_PLACE(idx), sym);
                        idx++;
                    } else {
                        uint32_t rep, val = 0;
                        if (sym == 16) {
                            if (idx == 0) return ITXI_E_CODES;
                            val = ITXI_LEN_GET(T, ITXI_PLACE(idx - 1));
                            rep = 3u + itxi_bits(in, 2);
                        } else if (sym == 17) {
                            rep = 3u + itxi_bits(in, 3);
                        } else {
                            rep = 11u + itxi_bits(in, 7);
                        }
                        if (idx + rep > nl + nd) return ITXI_E_CODES;
                        for (uint32_t k = 0; k < rep; k++) ITXI_LEN_SET(T, ITXI_PLACE(idx + k), val);
                        idx += rep;
                    }
                    if (itxi_overrun(in)) return ITXI_E_INPUT;
                }
#undef ITXI_PLACE
                if (ITXI_LEN_GET(T, 256u) == 0) return ITXI_E_CODES;                             // no end-of-block code
            }
            // The lengths leave the symbol table's room before the symbols move in: 40 words into the gap of the block's scratch region
            // between its literals (behind the word the next literals complete) and its tokens — literals + 4 x tokens stay below
            // 87 382 (ITXI_REGION's comment), the words end at most 179 bytes past the literals: 2.5 KB to spare — and are read back
            // eight lengths to a word. The next literals write over them.
            uint32_t *stage = lit32 + (((n_lit + 4u + 15u) & ~15u) >> 2);
            for (uint32_t w = 0; w < ITXI_STAGE_WORDS; w++) stage[w] = itxi_lens_word(T, ln, nullptr, w);
            uint32_t n01;
            int32_t left = itxi_construct(T, ln, lc, false, T.loffs, stage, 0, nl, n01);
            if (type == 2 && left != 0 && (left < 0 || nl != n01)) return ITXI_E_CODES;          // the fixed code is incomplete by definition
            left = itxi_construct(T, ln, dc, true, T.doffs, stage, 288, nd, n01);
            if (type == 2 && left != 0 && (left < 0 || nd != n01)) return ITXI_E_CODES;

            // A turn: up to ITXI_LITS literal/length codes — the first that is no literal ends them — then, behind a length, its
            // distance code. A turn costs the wave what its instructions cost whatever a lane finds in it (64 blocks in 64 states:
            // every branch is taken by some lane), and its fixed part outweighs a decode: most codes are literals, so a second and
            // third try at one takes turns off every block for a fraction of a turn each.
            for (;;) {
                uint32_t cl, at, sym;
#define ITXI_LITLEN()                                                                                                       \
    itxi_refill(in);                                                                                                        \
    at = itxi_decode(lc, T.loffs, ln, (uint32_t)in.bb & 0x7fffu, cl);                                                       \
    if (cl == 0) return ITXI_E_SYMBOL;                                                                                      \
    itxi_bits(in, cl);                                                                                                      \
    sym = ITXI_LSYM_GET(T, at)
#define ITXI_LITERAL()                                                                                                      \
    if (itxi_past(in)) return ITXI_E_INPUT; /* literals out of the bytes behind the block: stop before the padding ends */ \
    if (produced >= usize) return ITXI_E_OUTPUT;                                                                            \
    ITXI_PUT_LIT(sym);                                                                                                      \
    run++;                                                                                                                  \
    produced++
                ITXI_LITLEN();
#pragma unroll
                for (int more = 1; more < ITXI_LITS; more++) {
                    // ... when at least half of the wave's lanes would use it (blocks that are nearly all matches: legacy content)
                    if (2u * (uint32_t)__builtin_popcountll(ITXI_BALLOT(sym < 256u)) < (uint32_t)__builtin_popcountll(ITXI_BALLOT(true))) break;
                    if (sym >= 256u) break;
                    ITXI_LITERAL();
                    ITXI_LITLEN();
                }
                if (sym < 256u) {
                    ITXI_LITERAL();
                    continue;
                }
#undef ITXI_LITLEN
#undef ITXI_LITERAL
                if (sym == 256u) break;
                const uint32_t li = sym - 257u;
                if (li >= 29u) return ITXI_E_SYMBOL;
                // RFC 1951 3.2.5 as arithmetic (a table lookup here is a dependent global load per lane: measured 1 us each):
                // length codes 257..264 stand for 3..10; then groups of four share 1, 2, .. 5 extra bits; 285 is 258
                const uint32_t le = (li < 8u || li == 28u) ? 0u : (li - 4u) >> 2;
                const uint32_t lb = li < 8u ? 3u + li : li == 28u ? 258u : 3u + ((4u + (li & 3u)) << le);
                const uint32_t len = lb + itxi_bits(in, le);
                itxi_refill(in);
                at = itxi_decode(dc, T.doffs, ln, (uint32_t)in.bb & 0x7fffu, cl);
                if (cl == 0) return ITXI_E_SYMBOL;
                itxi_bits(in, cl);
                const uint32_t di = ITXI_AT(T.dsym, at);
                if (di >= 30u) return ITXI_E_SYMBOL;
                // distance codes 0..3 stand for 1..4; then pairs share 1, 2, .. 13 extra bits
                const uint32_t de = di < 4u ? 0u : (di - 2u) >> 1;
                const uint32_t db = di < 4u ? 1u + di : 1u + ((2u + (di & 1u)) << de);
                const uint32_t dist = db + itxi_bits(in, de);
                if (itxi_overrun(in)) return ITXI_E_INPUT;
                if (dist > produced) return ITXI_E_DIST;
                if (len > usize - produced) return ITXI_E_OUTPUT;
                if (run > 255u) {                                  // (rare: the literals of one token reach beyond its 8 bits)
                    ITXI_EXP_STORE(*(K.tok_top - 1 - n_tok) = ITXI_TOK_RUN | run);
                    n_tok++;
                    run = 0;
                }
                ITXI_EXP_STORE(*(K.tok_top - 1 - n_tok) = (len - 3u) | ((dist - 1u) << 8) | (run << 23));
                n_tok++;
                run = 0;
                produced += len;
            }
        }
        if (itxi_overrun(in)) return ITXI_E_INPUT;
        if (last) break;
    }
    if (produced != usize) return ITXI_E_OUTPUT;
    if (n_lit & 3u) lit32[n_lit >> 2] = lacc;                      // the scratch has room past the last literal
#undef ITXI_PUT_LIT
    K.n_lit = n_lit;
    K.n_tok = n_tok;
    return ITXI_OK;
}

// ---------------------------------------------------------------------------------------------------- pass 2: bytes

struct ItxiOut {
    uint8_t *g;                            // the whole output buffer
    uint32_t g0, gend;                     // this block's bytes are g[g0, gend)
    uint32_t gp;                           // next byte to produce
    uint32_t fl;                           // bytes below this offset are written back
};

// write back ring bytes [fl, to): whole aligned stripes as words, ragged ends (a block's first and last bytes share
// words with its neighbours, which other waves write) byte by byte
ITXI_FN void itxi_writeback(const uint32_t *ring32, ItxiOut &o, uint32_t to, uint32_t lane)
{
    const uint8_t *ring8 = reinterpret_cast<const uint8_t *>(ring32);
    uint32_t from = o.fl;
    while (from < to) {
        const uint32_t stop = (from | (ITXI_STRIPE - 1u)) + 1u;                    // end of the stripe `from` lies in
        const uint32_t upto = stop < to ? stop : to;
        if ((from & (ITXI_STRIPE - 1u)) == 0 && upto == stop) {
            for (uint32_t k = lane; k < ITXI_STRIPE / 4; k += ITXI_WAVE)
                reinterpret_cast<uint32_t *>(o.g)[(from >> 2) + k] = ring32[((from & ITXI_MASK) >> 2) + k];
        } else {
            for (uint32_t a = from + lane; a < upto; a += ITXI_WAVE) o.g[a] = ring8[a & ITXI_MASK];
        }
        from = upto;
    }
    o.fl = to;
}

// after gp moved: every stripe that is complete leaves
ITXI_FN void itxi_flush_full(const uint32_t *ring32, ItxiOut &o, uint32_t lane)
{
    const uint32_t full = o.gp & ~(ITXI_STRIPE - 1u);
    if (full > o.fl) itxi_writeback(ring32, o, full, lane);
}

#ifndef ITXI_LSTAGE
#define ITXI_LSTAGE 2048u                  // pass 2: literal bytes staged in LDS per refill (16 bytes per lane and load)
#endif

struct ItxiLit {
    const uint8_t *lit;                    // the block's literal string (16-byte aligned)
    uint32_t *stage32;                     // LDS, ITXI_LSTAGE bytes
    uint32_t base, n_lit;                  // stage32 holds lit[base, base + ITXI_LSTAGE) as far as it exists; base % 16 == 0
};

// the stage refilled so that it starts at (the 16-byte line of) literal `at`; at < n_lit: there is something to stage
ITXI_FN void itxi_stage_fill(ItxiLit &L, uint32_t at, uint32_t lane)
{
    L.base = at & ~15u;
    uint32_t have = L.n_lit - L.base;
    if (have > ITXI_LSTAGE) have = ITXI_LSTAGE;
    for (uint32_t k = lane * 16u; k < have; k += ITXI_WAVE * 16u) {                // reads up to 15 bytes past n_lit: inside the block's scratch
        const uint32_t *src = reinterpret_cast<const uint32_t *>(L.lit + L.base + k);
        const uint32_t a = ITXI_LOADW(src, 0), b = ITXI_LOADW(src, 1), c = ITXI_LOADW(src, 2), d = ITXI_LOADW(src, 3);
        L.stage32[k / 4] = a;
        L.stage32[k / 4 + 1] = b;
        L.stage32[k / 4 + 2] = c;
        L.stage32[k / 4 + 3] = d;
    }
}

// n literal bytes lit[lp, lp + n) into the ring, a stripe's worth at a time so the ring never laps itself. The literals
// come through an LDS stage refilled 4 KiB at a time: one global round trip per few hundred tokens instead of one each.
ITXI_FN void itxi_literals(uint32_t *ring32, ItxiOut &o, ItxiLit &L, uint32_t lp, uint32_t n, uint32_t lane)
{
    uint8_t *ring8 = reinterpret_cast<uint8_t *>(ring32);
    const uint8_t *stage8 = reinterpret_cast<const uint8_t *>(L.stage32);
    uint32_t done = 0;
    while (done < n) {
        uint32_t m = n - done;
        const uint32_t room = ITXI_STRIPE - (o.gp & (ITXI_STRIPE - 1u));
        if (m > room) m = room;
        const uint32_t at = lp + done;
        if (at < L.base || at + m > L.base + ITXI_LSTAGE) itxi_stage_fill(L, at, lane);
        for (uint32_t k = lane; k < m; k += ITXI_WAVE) ring8[(o.gp + k) & ITXI_MASK] = stage8[at - L.base + k];
        o.gp += m;
        done += m;
        itxi_flush_full(ring32, o, lane);
    }
}

// Replays a block's tokens (validated by pass 1: distances inside the block, lengths inside usize) into out[g0, g0 + usize).
// Tokens are fetched a wave's width at a time, one per lane (lane j = token j of the batch), the next batch while the current
// one is replayed. stage32 must lie right behind the ring (ring32 + ITXI_RING / 4); bmap32: (ITXI_BMAP + 64) / 32 words.
//
// Per batch, in the vector pipes: where every token's bytes go and where its literals come from (prefix sums). Then the batch
// is replayed a WAVE OF BYTES at a time, whatever the tokens' sizes: lane q of a step produces output byte base + q — it finds
// its token (a bitmap of token starts in LDS, a ballot, a bit count), reads that token's fields from the lane that holds them,
// and takes its byte from the literal stage or, for a match, from the ring (the few matches that reach further back: from what
// was written back). A step may only read what earlier steps wrote: it ends in front of the first byte whose source lies inside
// the step itself (a match right behind its own source; a match that overlaps itself reads its first `distance` bytes again,
// so it too ends a step at most once). With one token per step — what this loop used to do, a token of 9 bytes on average —
// most lanes of most steps had nothing to do, and a literal-heavy stream (real qualities: 3 - 4 literals per match) replayed at
// half the rate of a match-heavy one.
// Tokens the steps do not take (literals that do not fit the stage in one piece, a token behind the bitmap's reach, damaged
// values) go the long way round one at a time, as before.
ITXI_FN int itxi_resolve(uint32_t *ring32, uint32_t *stage32, uint32_t *bmap32, const uint8_t *lit, const uint32_t *tok_top, uint32_t n_lit, uint32_t n_tok, uint8_t *out,
                         uint32_t g0, uint32_t usize, uint32_t lane)
{
    uint8_t *ring8 = reinterpret_cast<uint8_t *>(ring32);
    ItxiOut o;
    o.g = out;
    o.g0 = g0;
    o.gend = g0 + usize;
    o.gp = g0;
    o.fl = g0;
    ItxiLit L;
    L.lit = lit;
    L.stage32 = stage32;
    L.base = 0xfffff000u;                                          // nothing staged yet
    L.n_lit = n_lit;
    uint32_t lp = 0;
    if (n_lit > usize || n_tok > ITXI_MAX_TOK) return ITXI_E_OUTPUT;
    if (reinterpret_cast<const uint8_t *>(stage32) != ring8 + ITXI_RING) return ITXI_E_OUTPUT;      // a step reads both through ring8
    uint32_t nw = 0;
    if (lane < n_tok) nw = ITXI_LOADW(tok_top - 1 - lane, 0);
    for (uint32_t t0 = 0; t0 < n_tok; t0 += ITXI_WAVE) {
        const uint32_t cw = nw;
        const uint32_t tn = t0 + ITXI_WAVE + lane;
        if (tn < n_tok) nw = ITXI_LOADW(tok_top - 1 - tn, 0);
        const uint32_t nb = n_tok - t0 < ITXI_WAVE ? n_tok - t0 : ITXI_WAVE;
        const bool have = lane < nb;
        const bool is_run = (cw & ITXI_TOK_RUN) != 0;
        const uint32_t v_run = !have ? 0u : is_run ? cw & ~ITXI_TOK_RUN : (cw >> 23) & 0xffu;
        const uint32_t v_len = (have && !is_run) ? (cw & 0xffu) + 3u : 0u;
        const uint32_t cd = (have && !is_run) ? ((cw >> 8) & 0x7fffu) + 1u : 0u;
        if (ITXI_BALLOT(v_run > ITXI_MAX_BLOCK) != 0) return ITXI_E_OUTPUT;        // (keeps the sums below inside 32 bits whatever the words hold)
        const uint32_t v_tot = v_run + v_len;
        const uint32_t e_gp = ITXI_SCAN_ADD(v_tot, lane), e_lp = ITXI_SCAN_ADD(v_run, lane);           // inclusive
        const uint32_t v_st = e_gp - v_tot, v_lp = lp + e_lp - v_run;                                  // start of the token's bytes in the batch; of its literals
        const uint32_t lit_total = ITXI_BCAST(e_lp, ITXI_WAVE - 1u), B = ITXI_BCAST(e_gp, ITXI_WAVE - 1u);
        const uint32_t gp0 = o.gp;
        if (lit_total > n_lit - lp || B > o.gend - gp0) return ITXI_E_OUTPUT;
        // the batch's literals staged all at once when they fit (they nearly always do): no refill inside the batch
        const bool staged = lit_total <= ITXI_LSTAGE - 16u;
        if (staged && lit_total && (lp < L.base || lp + lit_total > L.base + ITXI_LSTAGE)) itxi_stage_fill(L, lp, lane);
        const uint32_t v_ls = v_lp - L.base;                       // the literals' place in the stage (meaningful when staged)
        // a token the steps may take: its literals staged, its bytes inside the bitmap's reach, its match inside the block
        const bool v_ok = have && staged && v_tot != 0u && e_gp <= ITXI_BMAP && (is_run || cd <= gp0 - o.g0 + v_st + v_run);
        const uint64_t fastmask = ITXI_BALLOT(v_ok);
        if (fastmask != 0) {
            for (uint32_t k = lane; k < (ITXI_BMAP + 64u) / 32u; k += ITXI_WAVE) bmap32[k] = 0;
            if (v_ok) ITXI_LDS_OR(&bmap32[v_st >> 5], 1u << (v_st & 31u));
        }
        const uint32_t pk1 = v_run | (v_ls << 16);               // (staged: run and stage offset below 2^11)
        uint32_t j = 0;
        while (j < nb) {
            const uint64_t rest = ~fastmask >> j;
            uint32_t jn = rest ? j + (uint32_t)__builtin_ctzll(rest) : ITXI_WAVE;
            if (jn > nb) jn = nb;
            if (jn > j) {
                // tokens [j, jn) as steps of a wave of bytes
                const uint32_t seg_hi = ITXI_BCAST(e_gp, jn - 1u);
                uint32_t base = ITXI_BCAST(v_st, j), n_before = j;                 // tokens of the batch that start before `base`
                while (base < seg_hi) {
                    const uint32_t p = base + lane;
                    const bool valid = p < seg_hi;
                    const uint32_t sbit = valid ? (bmap32[p >> 5] >> (p & 31u)) & 1u : 0u;
                    const uint64_t M = ITXI_BALLOT(sbit != 0);
                    const uint32_t t = n_before + ITXI_MBCNT(M, lane) + sbit - 1u;
                    const uint32_t a_st = ITXI_LANE_READ(v_st, t), a1 = ITXI_LANE_READ(pk1, t), dist = ITXI_LANE_READ(cd, t);
                    const uint32_t off = p - a_st, run = a1 & 0xffffu, ls = a1 >> 16;
                    const bool is_lit = off < run;
                    const uint32_t k = off - run;                                  // byte of the match
                    uint32_t km = k;
                    const bool wraps = valid && !is_lit && k >= dist;              // the match overlaps itself: its first `dist` bytes again
                    if (ITXI_BALLOT(wraps) != 0) km = wraps ? k % dist : k;
                    const int32_t src = (int32_t)(a_st + run + km) - (int32_t)dist;        // in the batch's coordinates; before the batch: negative
                    const uint64_t cf = ITXI_BALLOT(valid && !is_lit && src >= (int32_t)base);
                    uint32_t n_step = seg_hi - base < ITXI_WAVE ? seg_hi - base : ITXI_WAVE;
                    if (cf != 0) {
                        const uint32_t first = (uint32_t)__builtin_ctzll(cf);     // never lane 0: its source lies before its own place
                        if (first < n_step) n_step = first;
                    }
                    const bool act = lane < n_step;
                    const bool far = act && !is_lit && dist > ITXI_NEAR;
                    const uint32_t a_lit = ITXI_RING + ls + off, a_ring = (gp0 + (uint32_t)src) & ITXI_MASK;
                    uint32_t from = is_lit ? a_lit : a_ring;
                    if (!act || far) from = 0;
                    uint8_t byte = ring8[from];
                    if (ITXI_BALLOT(far) != 0) {
                        // further back than the ring reaches: those bytes left in whole stripes long ago (a statement of its own,
                        // not a selected pointer: that would be a flat load)
                        ITXI_FENCE();
                        if (far) byte = ITXI_LOADB(o.g, gp0 + (uint32_t)src);
                    }
                    if (act) ring8[(gp0 + p) & ITXI_MASK] = byte;
                    const uint64_t took = n_step >= 64u ? ~0ull : ((1ull << n_step) - 1ull);
                    n_before += (uint32_t)__builtin_popcountll(M & took);
                    base += n_step;
                    o.gp = gp0 + base;
                    if (o.gp - o.fl >= ITXI_LAG) itxi_flush_full(ring32, o, lane);
                }
                itxi_flush_full(ring32, o, lane);
                j = jn;
                if (j >= nb) break;
            }
            // token j the long way round
            const uint32_t run = ITXI_BCAST(v_run, j), len = ITXI_BCAST(v_len, j), dist = ITXI_BCAST(cd, j);
            const uint32_t lpj = ITXI_BCAST(v_lp, j);              // == lp + the runs of the batch's tokens before this one
            // guards that hold for every token pass 1 lets through; they keep a damaged token array from leaving the block
            if (lpj > n_lit || run > n_lit - lpj || run + len > o.gend - o.gp) return ITXI_E_OUTPUT;
            if (run) itxi_literals(ring32, o, L, lpj, run, lane);
            if (len) {
                if (dist == 0 || dist > o.gp - o.g0) return ITXI_E_DIST;
                const uint32_t src0 = o.gp - dist;
                if (dist <= ITXI_NEAR) {
                    if (dist >= 64u || dist >= len) {             // 64: the widest step any build takes
                        // a step's sources were all written before the step (earlier steps or earlier tokens)
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
                    // further back than the ring reaches: those bytes left in whole stripes long ago (dist > 3 KiB > len)
                    ITXI_FENCE();
                    for (uint32_t k0 = 0; k0 < len; k0 += ITXI_WAVE) {
                        const uint32_t k = k0 + lane;
                        if (k < len) ring8[(o.gp + k) & ITXI_MASK] = ITXI_LOADB(o.g, src0 + k);
                    }
                }
                o.gp += len;
                itxi_flush_full(ring32, o, lane);
            }
            j++;
        }
        if (o.gp != gp0 + B) return ITXI_E_OUTPUT;
        lp += lit_total;
    }
    const uint32_t tail = n_lit - lp;
    if (tail != o.gend - o.gp) return ITXI_E_OUTPUT;
    if (tail) itxi_literals(ring32, o, L, lp, tail, lane);
    itxi_writeback(ring32, o, o.gend, lane);
    return ITXI_OK;
}
