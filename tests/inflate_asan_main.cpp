// Sanitizer harness of the device DEFLATE decoder's host build (tests/inflate_host.cpp): test infrastructure only.
// Reads cases from a file — u32 n_cases, then per case: u32 at, u32 comp_len, u32 usize, comp bytes — and runs each in
// EXACT-size heap buffers, so that AddressSanitizer sees any access outside what the callers promise the decoder:
//   input  : `at` bytes of offset + the compressed bytes + 24 bytes (a BGZF block's 8-byte trailer + 16 bytes of padding)
//   output : usize bytes
// Prints one line per case: rc, and a 64-bit FNV hash of the output when rc == 0.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

extern "C" int itx_inflate_host(const uint32_t *comp_words, uint32_t data_pos, uint32_t data_end, uint8_t *out, uint32_t g0, uint32_t usize,
                                uint32_t *n_lit, uint32_t *n_tok);

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    uint32_t n = 0;
    if (fread(&n, 4, 1, f) != 1) return 2;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t h[3];
        if (fread(h, 4, 3, f) != 3) return 2;
        const uint32_t at = h[0], clen = h[1], usize = h[2];
        const size_t in_bytes = (size_t)at + clen + 24;
        uint8_t *in = (uint8_t *)aligned_alloc(4, (in_bytes + 3) & ~(size_t)3);       // rounded to whole words: the decoder loads words
        memset(in, 0, (in_bytes + 3) & ~(size_t)3);
        if (clen && fread(in + at, 1, clen, f) != clen) return 2;
        uint8_t *out = (uint8_t *)malloc(usize ? usize : 1);
        const int rc = itx_inflate_host((const uint32_t *)in, at, at + clen, out, 0, usize, nullptr, nullptr);
        uint64_t fnv = 1469598103934665603ull;
        if (rc == 0)
            for (uint32_t k = 0; k < usize; k++) fnv = (fnv ^ out[k]) * 1099511628211ull;
        printf("%d %llu\n", rc, (unsigned long long)(rc == 0 ? fnv : 0));
        free(in);
        free(out);
    }
    fclose(f);
    return 0;
}
