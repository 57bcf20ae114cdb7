// Host build of the wave's DEFLATE decoder (iteres_amd/csrc/itx_inflate_core.h) with a one-lane "wave": test
// infrastructure only — the CPU suite fuzzes the decoder's logic against zlib here before it ever runs on a GPU.
#include <stdint.h>
#define ITXI_WAVE 1u
#define ITXI_FN static inline
#define ITXI_UNI(x) (x)
#define ITXI_LOADW(w, i) ((w)[i])
#define ITXI_LOADB(p, i) ((p)[i])
#define ITXI_FENCE() ((void)0)
#include "../iteres_amd/csrc/itx_inflate_core.h"

extern "C" int itx_inflate_host(const uint32_t *comp_words, uint32_t data_pos, uint32_t data_end, uint8_t *out, uint32_t g0, uint32_t usize)
{
    static thread_local ItxiLds S;
    return itxi_block(S, comp_words, data_pos, data_end, out, g0, usize, 0);
}
