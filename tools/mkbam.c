/* mkbam.c — fast generator of a synthetic coordinate-sorted BAM for end-to-end throughput runs (test tool).
 *
 *   mkbam <chrom.sizes> <n_reads> <out.bam> [seq_len=0] [seed=1] [xa_permille=0]
 *
 * Single-end reads, positions increasing along every chromosome (one read per mean gap, jittered), 50 % reverse strand,
 * MAPQ from {0,0,3,20,37,37,37,60}, read length 100-150, CIGAR nM; with seq_len > 0 every record carries that many bases
 * (4-bit codes of A/C/G/T) + qualities — what real BAMs look like: ~5x more bytes to inflate per record. xa_permille: that
 * many reads per thousand carry `XA:Z:<chrom>,<+-pos>,<len>M,<nm>;...` (1-3 alternatives anywhere in the genome) and NM:i.
 *
 * The file is a function of the arguments alone: reads are generated in segments of SEG reads, each from a generator seeded
 * by (seed, segment number), each compressed into its own run of BGZF blocks (level 1; libdeflate when the system has it,
 * zlib otherwise — MKBAM_ZLIB=1 forces zlib, MKBAM_LEVEL=n another level), by as many threads as OpenMP gives; segments are written in order. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

typedef struct { uint64_t s; } rng_t;
static inline uint64_t rnd(rng_t *g)
{
    uint64_t z = (g->s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

typedef struct {
    uint8_t *p;
    size_t n, cap;
} buf_t;
static inline void need(buf_t *b, size_t k)
{
    if (b->n + k > b->cap) {
        b->cap = (b->n + k) * 2 + 4096;
        b->p = realloc(b->p, b->cap);
        if (!b->p) abort();
    }
}
static inline void put(buf_t *b, const void *src, size_t k)
{
    need(b, k);
    memcpy(b->p + b->n, src, k);
    b->n += k;
}
static inline void put32(buf_t *b, uint32_t v) { put(b, &v, 4); }

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

/* ---- raw DEFLATE of one block, level 1 */
typedef struct libdeflate_compressor ld_comp;
static ld_comp *(*ld_alloc)(int);
static size_t (*ld_compress)(ld_comp *, const void *, size_t, void *, size_t);
static int use_ld, z_level = 1;                 /* MKBAM_LEVEL=1..9 (default 1; htslib writes level 6 with zlib) */
static void ld_probe(void)
{
    const char *lv = getenv("MKBAM_LEVEL");
    if (lv && atoi(lv) >= 1 && atoi(lv) <= 9) z_level = atoi(lv);
    if (getenv("MKBAM_ZLIB")) return;
    void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    ld_alloc = (ld_comp * (*)(int)) dlsym(h, "libdeflate_alloc_compressor");
    ld_compress = (size_t(*)(ld_comp *, const void *, size_t, void *, size_t))dlsym(h, "libdeflate_deflate_compress");
    use_ld = ld_alloc && ld_compress;
}

#define BLK 0xff00
static size_t bgzf_compress(const uint8_t *src, size_t n, uint8_t *dst)
{
    static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    size_t clen = 0;
    if (use_ld) {
        static __thread ld_comp *c;
        if (!c) c = ld_alloc(z_level);
        clen = c ? ld_compress(c, src, n, dst + 18, 0x10000) : 0;
    }
    if (!clen) {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        deflateInit2(&zs, z_level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = (Bytef *)src;
        zs.avail_in = (uInt)n;
        zs.next_out = dst + 18;
        zs.avail_out = 0x10000;
        deflate(&zs, Z_FINISH);
        clen = zs.total_out;
        deflateEnd(&zs);
    }
    memcpy(dst, hdr, 16);
    const uint16_t bsize = (uint16_t)(clen + 25);
    memcpy(dst + 16, &bsize, 2);
    const uint32_t crc = (uint32_t)crc32(crc32(0, NULL, 0), src, (uInt)n), isz = (uint32_t)n;
    memcpy(dst + 18 + clen, &crc, 4);
    memcpy(dst + 22 + clen, &isz, 4);
    return clen + 26;
}

/* the inflated bytes of `b` as BGZF blocks of BLK bytes (the last one shorter) appended to `out` */
static void compress_all(const buf_t *b, buf_t *out)
{
    for (size_t off = 0; off < b->n; off += BLK) {
        const size_t k = b->n - off < BLK ? b->n - off : BLK;
        need(out, 0x10100);
        out->n += bgzf_compress(b->p + off, k, out->p + out->n);
    }
}

static inline int put_dec(char *dst, long long v)          /* decimal digits, returns their number */
{
    char tmp[24];
    int k = 0;
    do {
        tmp[k++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    for (int i = 0; i < k; i++) dst[i] = tmp[k - 1 - i];
    return k;
}

#define SEG 65536
typedef struct {
    int chrom;
    long long first, count, serial;      /* reads [first, first + count) of the chromosome; serial = number of the first among all reads */
} seg_t;

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: mkbam <chrom.sizes> <n_reads> <out.bam> [seq_len=0] [seed=1] [xa_permille=0]\n");
        return 1;
    }
    const long long n_reads = atoll(argv[2]);
    const int seq_len = argc > 4 ? atoi(argv[4]) : 0;
    const uint64_t seed = argc > 5 ? strtoull(argv[5], 0, 0) : 1;
    const unsigned xa_pm = argc > 6 ? (unsigned)atoi(argv[6]) : 0;
    static char names[256][64];
    static long long sizes[256], nr_of[256];
    int nc = 0;
    FILE *cf = fopen(argv[1], "r");
    if (!cf) return 2;
    while (nc < 256 && fscanf(cf, "%63s %lld", names[nc], &sizes[nc]) == 2) nc++;
    fclose(cf);
    if (nc == 0) return 2;
    long long genome = 0;
    for (int c = 0; c < nc; c++) genome += sizes[c];
    FILE *f = fopen(argv[3], "wb");
    if (!f) return 3;
    ld_probe();

    /* header: its own blocks */
    {
        buf_t b = {0}, o = {0};
        char *text = malloc(64 + (size_t)nc * 128);
        int tl = sprintf(text, "@HD\tVN:1.0\tSO:coordinate\n");
        for (int c = 0; c < nc; c++) tl += sprintf(text + tl, "@SQ\tSN:%s\tLN:%lld\n", names[c], sizes[c]);
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
        compress_all(&b, &o);
        fwrite(o.p, 1, o.n, f);
        free(b.p);
        free(o.p);
        free(text);
    }

    /* the segments */
    long long done = 0, n_seg = 0;
    for (int c = 0; c < nc; c++) {
        nr_of[c] = c == nc - 1 ? n_reads - done : (long long)((double)n_reads * (double)sizes[c] / (double)genome);
        if (nr_of[c] < 0) nr_of[c] = 0;
        done += nr_of[c];
        n_seg += (nr_of[c] + SEG - 1) / SEG;
    }
    seg_t *seg = malloc(sizeof *seg * (size_t)(n_seg + 1));
    {
        long long k = 0, serial = 0;
        for (int c = 0; c < nc; c++)
            for (long long a = 0; a < nr_of[c]; a += SEG) {
                seg[k].chrom = c;
                seg[k].first = a;
                seg[k].count = nr_of[c] - a < SEG ? nr_of[c] - a : SEG;
                seg[k].serial = serial;
                serial += seg[k].count;
                k++;
            }
    }
    static const uint8_t mq[8] = {0, 0, 3, 20, 37, 37, 37, 60};
    static const uint8_t base2[4] = {0x11, 0x22, 0x44, 0x88};
    int io_error = 0;
#pragma omp parallel
    {
        buf_t b = {0}, o = {0};
        uint8_t *sq = malloc((size_t)seq_len * 2 + 64);
#pragma omp for ordered schedule(dynamic, 1)
        for (long long s = 0; s < n_seg; s++) {
            const seg_t *g = &seg[s];
            rng_t R = {seed * 0x2545f4914f6cdd1dull + (uint64_t)s * 0x9e3779b97f4a7c15ull + 1};
            const int c = g->chrom;
            const double mean_gap = (double)sizes[c] / (double)(nr_of[c] + 1);
            b.n = 0;
            o.n = 0;
            for (long long i = 0; i < g->count; i++) {
                const uint64_t r = rnd(&R);
                /* read k of the chromosome sits in [k, k + 1) mean gaps: increasing in k without any running state */
                int pos = (int)(((double)(g->first + i) + (double)(r >> 40) / (double)(1 << 24)) * mean_gap);
                if (pos >= sizes[c]) pos = (int)sizes[c] - 1;
                const int rl = 100 + (int)((r >> 8) % 51);
                const int flag = (r & 1) ? 16 : 0;
                char qn[32];
                qn[0] = 'r';
                int ql = 1 + put_dec(qn + 1, g->serial + i);
                qn[ql++] = 0;
                const int l_seq = seq_len;
                char xa[256];
                int xl = 0;
                if (xa_pm && (unsigned)((r >> 20) % 1000) < xa_pm) {
                    const uint64_t x = rnd(&R);
                    const int n_alt = 1 + (int)(x % 3);
                    xa[xl++] = 'X', xa[xl++] = 'A', xa[xl++] = 'Z';
                    for (int a = 0; a < n_alt; a++) {
                        const uint64_t y = rnd(&R);
                        const int ac = (int)(y % (uint64_t)nc);
                        const long long ap = 1 + (long long)((y >> 16) % (uint64_t)(sizes[ac] > rl ? sizes[ac] - rl : 1));
                        xl += sprintf(xa + xl, "%s,%c", names[ac], (y >> 8) & 1 ? '-' : '+');
                        xl += put_dec(xa + xl, ap);
                        xl += sprintf(xa + xl, ",%dM,%d;", rl, (int)((y >> 9) & 3));
                    }
                    xa[xl++] = 0;
                    xa[xl++] = 'N', xa[xl++] = 'M', xa[xl++] = 'C', xa[xl++] = (char)((x >> 8) & 3);
                }
                const uint32_t block = 32 + (uint32_t)ql + 4 + (uint32_t)((l_seq + 1) / 2 + l_seq) + (uint32_t)xl;
                need(&b, block + 4);
                uint32_t *w = (uint32_t *)(void *)(b.p + b.n);          /* unaligned stores are fine on the hosts this runs on */
                uint32_t hdr[9] = {block,
                                   (uint32_t)c,
                                   (uint32_t)pos,
                                   ((uint32_t)reg2bin(pos, pos + rl) << 16) | ((uint32_t)mq[(r >> 4) & 7] << 8) | (uint32_t)ql,
                                   ((uint32_t)flag << 16) | 1u,
                                   (uint32_t)l_seq,
                                   0xffffffffu,
                                   0xffffffffu,
                                   0};
                memcpy(w, hdr, 36);
                b.n += 36;
                memcpy(b.p + b.n, qn, (size_t)ql);
                b.n += (size_t)ql;
                const uint32_t cg = ((uint32_t)rl << 4) | 0u;
                memcpy(b.p + b.n, &cg, 4);
                b.n += 4;
                if (l_seq) {
                    const int nb = (l_seq + 1) / 2;
                    for (int k = 0; k < nb; k += 32) {                /* 2 random bits per byte: both nibbles the same base code */
                        uint64_t y = rnd(&R);
                        for (int j = 0; j < 32; j++, y >>= 2) sq[k + j] = base2[y & 3];
                    }
                    uint8_t *q = sq + nb + 32;
                    for (int k = 0; k < l_seq; k++) q[k] = (uint8_t)(20 + ((r >> (k & 31)) & 15));
                    memcpy(b.p + b.n, sq, (size_t)nb);
                    memcpy(b.p + b.n + nb, q, (size_t)l_seq);
                    b.n += (size_t)(nb + l_seq);
                }
                if (xl) {
                    memcpy(b.p + b.n, xa, (size_t)xl);
                    b.n += (size_t)xl;
                }
            }
            compress_all(&b, &o);
#pragma omp ordered
            {
                if (fwrite(o.p, 1, o.n, f) != o.n) io_error = 1;
            }
        }
        free(b.p);
        free(o.p);
        free(sq);
    }
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    fwrite(eof, 1, 28, f);
    if (fclose(f) != 0 || io_error) {
        fprintf(stderr, "mkbam: write error\n");
        return 4;
    }
    free(seg);
    fprintf(stderr, "wrote %lld records (%s level %d)\n", done, use_ld ? "libdeflate" : "zlib", z_level);
    return 0;
}
