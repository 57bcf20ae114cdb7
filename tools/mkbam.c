/* mkbam.c — fast generator of a synthetic coordinate-sorted BAM for end-to-end throughput runs (test tool).
 *
 *   mkbam <chrom.sizes> <n_reads> <out.bam> [seq_len=0] [seed=1]
 *
 * Single-end reads, uniform positions (exponential gaps), 50 % reverse strand, MAPQ from {0,0,3,20,37,37,37,60},
 * read length 100-150, CIGAR nM; with seq_len > 0 every record carries that many bases + qualities (what real
 * BAMs look like: ~5x more bytes to inflate per record). BGZF blocks are compressed in parallel (OpenMP, level 1). */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

static uint64_t rng_state;
static inline uint64_t rnd(void)
{
    uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

typedef struct {
    uint8_t *p;
    size_t n, cap;
} buf_t;
static void put(buf_t *b, const void *src, size_t k)
{
    if (b->n + k > b->cap) {
        b->cap = (b->n + k) * 2 + 4096;
        b->p = realloc(b->p, b->cap);
    }
    memcpy(b->p + b->n, src, k);
    b->n += k;
}
static void put32(buf_t *b, uint32_t v) { put(b, &v, 4); }

static int reg2bin(int beg, int end)
{
    --end;
    if (beg >> 14 == end >> 14) return ((1 << 15) - 1) / 7 + (beg >> 14);
    if (beg >> 17 == end >> 17) return ((1 << 12) - 1) / 7 + (beg >> 17);
    if (beg >> 20 == end >> 20) return ((1 << 9) - 1) / 7 + (beg >> 20);
    if (beg >> 23 == end >> 23) return ((1 << 6) - 1) / 7 + (beg >> 23);
    if (beg >> 26 == end >> 26) return ((1 << 3) - 1) / 7 + (beg >> 26);
    return 0;
}

#define BLK 0xff00
static size_t bgzf_compress(const uint8_t *src, size_t n, uint8_t *dst)
{
    static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = (Bytef *)src;
    zs.avail_in = (uInt)n;
    zs.next_out = dst + 18;
    zs.avail_out = 0x10000;
    deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    memcpy(dst, hdr, 16);
    const uint16_t bsize = (uint16_t)(clen + 25);
    memcpy(dst + 16, &bsize, 2);
    const uint32_t crc = (uint32_t)crc32(crc32(0, NULL, 0), src, (uInt)n), isz = (uint32_t)n;
    memcpy(dst + 18 + clen, &crc, 4);
    memcpy(dst + 22 + clen, &isz, 4);
    return clen + 26;
}

static void flush_blocks(FILE *f, buf_t *b, int final)
{
    const size_t nb = final ? (b->n + BLK - 1) / BLK : b->n / BLK;
    if (!nb) return;
    uint8_t *out = malloc(nb * 0x10100);
    size_t *len = malloc(nb * sizeof *len);
#pragma omp parallel for schedule(dynamic, 8)
    for (long i = 0; i < (long)nb; i++) {
        const size_t off = (size_t)i * BLK, k = b->n - off < BLK ? b->n - off : BLK;
        len[i] = bgzf_compress(b->p + off, k, out + (size_t)i * 0x10100);
    }
    for (size_t i = 0; i < nb; i++) fwrite(out + i * 0x10100, 1, len[i], f);
    const size_t used = nb * BLK < b->n ? nb * BLK : b->n;
    memmove(b->p, b->p + used, b->n - used);
    b->n -= used;
    free(out);
    free(len);
}

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: mkbam <chrom.sizes> <n_reads> <out.bam> [seq_len=0] [seed=1]\n");
        return 1;
    }
    const long long n_reads = atoll(argv[2]);
    const int seq_len = argc > 4 ? atoi(argv[4]) : 0;
    rng_state = argc > 5 ? strtoull(argv[5], 0, 0) : 1;
    char names[256][64];
    long long sizes[256];
    int nc = 0;
    FILE *cf = fopen(argv[1], "r");
    if (!cf) return 2;
    while (nc < 256 && fscanf(cf, "%63s %lld", names[nc], &sizes[nc]) == 2) nc++;
    fclose(cf);
    long long genome = 0;
    for (int c = 0; c < nc; c++) genome += sizes[c];
    FILE *f = fopen(argv[3], "wb");
    if (!f) return 3;
    buf_t b = {0};
    char text[16384];
    int tl = snprintf(text, sizeof text, "@HD\tVN:1.0\tSO:coordinate\n");
    for (int c = 0; c < nc; c++) tl += snprintf(text + tl, sizeof text - tl, "@SQ\tSN:%s\tLN:%lld\n", names[c], sizes[c]);
    put(&b, "BAM\1", 4);
    put32(&b, (uint32_t)tl);
    put(&b, text, (size_t)tl);
    put32(&b, (uint32_t)nc);
    for (int c = 0; c < nc; c++) {
        const uint32_t l = (uint32_t)strlen(names[c]) + 1;
        put32(&b, l);
        put(&b, names[c], l);
        put32(&b, (uint32_t)sizes[c]);
    }
    static const uint8_t mq[8] = {0, 0, 3, 20, 37, 37, 37, 60};
    uint8_t *seq = calloc((size_t)seq_len + 8, 2);
    long long done = 0;
    for (int c = 0; c < nc; c++) {
        const long long nr = c == nc - 1 ? n_reads - done : (long long)((double)n_reads * sizes[c] / genome);
        const double mean_gap = (double)sizes[c] / (double)(nr + 1);
        double p = 0;
        for (long long i = 0; i < nr; i++) {
            const uint64_t r = rnd();
            p += mean_gap * (0.25 + 1.5 * (double)(r >> 40) / (double)(1 << 24));       /* increasing, mean = mean_gap */
            int pos = (int)p;
            if (pos >= sizes[c]) pos = (int)sizes[c] - 1;
            const int rl = 100 + (int)((r >> 8) % 51);
            const int flag = (r & 1) ? 16 : 0;
            char qn[32];
            const int ql = snprintf(qn, sizeof qn, "r%lld", done + i) + 1;
            const int l_seq = seq_len;
            const uint32_t block = 32 + (uint32_t)ql + 4 + (uint32_t)((l_seq + 1) / 2 + l_seq);
            put32(&b, block);
            put32(&b, (uint32_t)c);
            put32(&b, (uint32_t)pos);
            put32(&b, ((uint32_t)reg2bin(pos, pos + rl) << 16) | ((uint32_t)mq[(r >> 4) & 7] << 8) | (uint32_t)ql);
            put32(&b, ((uint32_t)flag << 16) | 1u);
            put32(&b, (uint32_t)l_seq);
            put32(&b, 0xffffffffu);
            put32(&b, 0xffffffffu);
            put32(&b, 0);
            put(&b, qn, (size_t)ql);
            put32(&b, ((uint32_t)rl << 4) | 0u);
            if (l_seq) {
                for (int k = 0; k < (l_seq + 1) / 2; k++) seq[k] = (uint8_t)(0x11 << ((rnd() >> 13) & 3));
                for (int k = 0; k < l_seq; k++) seq[(l_seq + 1) / 2 + k] = (uint8_t)(20 + ((r >> (k & 31)) & 15));
                put(&b, seq, (size_t)((l_seq + 1) / 2 + l_seq));
            }
            if (b.n >= (size_t)BLK * 4096) flush_blocks(f, &b, 0);
        }
        done += nr;
    }
    flush_blocks(f, &b, 1);
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    fwrite(eof, 1, 28, f);
    fclose(f);
    fprintf(stderr, "wrote %lld records\n", done);
    return 0;
}
