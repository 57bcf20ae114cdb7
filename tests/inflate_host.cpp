// Host build of the device DEFLATE decoder (iteres_amd/csrc/itx_inflate_core.h) with a one-lane "wave" and plain
// arrays for the per-decoder tables: test infrastructure only — the CPU suite fuzzes both passes against zlib here
// before they ever run on a GPU.
#include <stdint.h>
#include <vector>
#define ITXI_WAVE 1u
// the input form is the build's choice: -DITXI_SIMPLE_IN (what the device build uses) or the 16-byte FIFO
#define ITXI_FN static inline
#define ITXI_UNI(x) (x)
#define ITXI_BCAST(v, j) (v)
#define ITXI_SCAN_ADD(v, lane) (v)
#define ITXI_LANE_READ(v, j) (v)
#define ITXI_BALLOT(p) ((uint64_t)((p) ? 1u : 0u))
#define ITXI_MBCNT(m, lane) 0u
#define ITXI_LDS_OR(ptr, v) ((void)(*(ptr) |= (v)))
static inline uint32_t itxi_bitrev32(uint32_t x)
{
    x = (x >> 16) | (x << 16);
    x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8);
    x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2);
    return ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1);
}
#define ITXI_BITREV32(x) itxi_bitrev32(x)
static inline uint32_t itxi_pksign16(uint32_t a, uint32_t b)
{
    const uint16_t lo = (uint16_t)((uint16_t)a - (uint16_t)b), hi = (uint16_t)((uint16_t)(a >> 16) - (uint16_t)(b >> 16));
    return (uint32_t)(lo >> 15) | (uint32_t)(hi >> 15) << 16;
}
#define ITXI_PKSIGN16(a, b) itxi_pksign16(a, b)
#define ITXI_AT(p, i) (p)[(i)]
#define ITXI_LOADW(w, i) ((w)[i])
#define ITXI_LOADB(p, i) ((p)[i])
#define ITXI_FENCE() ((void)0)
#include "../iteres_amd/csrc/itx_inflate_core.h"

extern "C" int itx_inflate_host(const uint32_t *comp_words, uint32_t data_pos, uint32_t data_end, uint8_t *out, uint32_t g0, uint32_t usize,
                                uint32_t *n_lit, uint32_t *n_tok)
{
    static thread_local uint16_t loffs[16], doffs[16];
#ifndef ITXI_SYM16
    static thread_local uint8_t lsym8[288], dsym[32];
    static thread_local uint32_t lhi[9];
#else
    static thread_local uint16_t lsym16[288];
    static thread_local uint8_t dsym[32];
#endif
    static thread_local uint32_t mem32[(ITXI_RING + ITXI_LSTAGE + ITXI_BMAP / 8 + 8) / 4];       // the literal stage right behind the ring (itxi_resolve's contract), then the bitmap
    uint32_t *ring32 = mem32, *stage32 = mem32 + ITXI_RING / 4, *bmap32 = mem32 + (ITXI_RING + ITXI_LSTAGE) / 4;
    // the block's scratch region: literals from its bottom, tokens from its top; guard words either side catch a writer that leaves it
    static thread_local std::vector<uint32_t> region32(ITXI_REGION / 4 + 8);
    region32[0] = region32[1] = region32[2] = region32[3] = 0xfeedc0deu;
    for (int k = 0; k < 4; k++) region32[ITXI_REGION / 4 + 4 + k] = 0xfeedc0deu;
    uint8_t *lit = reinterpret_cast<uint8_t *>(region32.data() + 4);
    uint32_t *tok_top = region32.data() + 4 + ITXI_REGION / 4;
    if (usize > ITXI_MAX_BLOCK) return ITXI_E_OUTPUT;
#ifndef ITXI_SYM16
    ItxiTab T{lsym8, lhi, dsym, loffs, doffs};
#else
    ItxiTab T{lsym16, dsym, loffs, doffs};
#endif
    ItxiTokens K{lit, tok_top, 0, 0};
    int rc = itxi_tokens(T, 0, comp_words, data_pos, data_end, usize, K);
    if (n_lit) *n_lit = K.n_lit;
    if (n_tok) *n_tok = K.n_tok;
    for (int k = 0; k < 4; k++)
        if (region32[k] != 0xfeedc0deu || region32[ITXI_REGION / 4 + 4 + k] != 0xfeedc0deu) return 100;           // pass 1 wrote outside its region
    if (rc == ITXI_OK && (uint64_t)K.n_lit + 4u * (uint64_t)K.n_tok + 3u > ITXI_REGION) return 101;                   // literals and tokens met
    if (rc != ITXI_OK) return rc;
    return itxi_resolve(ring32, stage32, bmap32, lit, tok_top, K.n_lit, K.n_tok, out, g0, usize, 0);
}
