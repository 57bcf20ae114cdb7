/* mkbam.c — fast generator of a synthetic coordinate-sorted BAM for end-to-end throughput runs (test tool).
 *
 *   mkbam <chrom.sizes> <n_reads> <out.bam> [seq_len=0] [seed=1] [xa_permille=0] [key=value ...]
 *
 * Positions increase along every chromosome (one read per mean gap, jittered), 50 % reverse strand, MAPQ from
 * {0,0,3,20,37,37,37,60}; with seq_len > 0 every record carries bases (4-bit codes) + qualities — what real BAMs look
 * like: ~5x more bytes to inflate per record. xa_permille: that many reads per thousand carry
 * `XA:Z:<chrom>,<+-pos>,<len>M,<nm>;...` (1-3 alternatives anywhere in the genome) and NM:i.
 *
 * key=value options (what the measured inputs have to look like):
 *   content=legacy   (default) round-2 bytes exactly: read length 100-150 whatever seq_len, both nibbles of a SEQ byte the
 *                    same base, qualities a sliding 4-bit window of one 64-bit word (period 32), names r<serial>. 97 % of the
 *                    inflated bytes come out of LZ77 matches — the kindest case for a DEFLATE decoder.
 *   content=hiseq    independent bases per nibble (0.2 % N), 40-value qualities (Phred 2..41: a slowly falling level,
 *                    per-base noise, '#' tails), Illumina-style names (instrument:run:lane:tile:x:y), read length = seq_len.
 *   content=novaseq  the same with 4-bin qualities {2,12,23,37} in runs.
 *   cigar=simple     (default) nM
 *   cigar=mixed      5 % of the reads: aS bM | aM dD bM | aM iI bM | aM nN bM (bam_calend's D / N arm, bam.c:17-27)
 *   paired=1         fragments of two reads (isize ~ N(350, 60), clamped to the read length .. 700; FR orientation, either
 *                    read first; flags 99/147/83/163), 2 % of the fragments with an unmapped mate (flag 0x8 / 0x4 records
 *                    at the mapped read's place) — generic.c:836-860. n_reads counts records.
 *   pileup=K         K loci of the genome at ~2000x depth: the reads of 1000 mean gaps either side land within 64 bp
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

enum { C_LEGACY, C_HISEQ, C_NOVASEQ };
static int content = C_LEGACY, cigar_mixed, paired, n_pileup;
static int nc;
static char names[256][64];
static long long sizes[256], nr_of[256];
static const uint8_t mq[8] = {0, 0, 3, 20, 37, 37, 37, 60};

/* ---- sequence + quality content */
static void fill_seq_legacy(rng_t *R, uint64_t r, int l_seq, uint8_t *sq, uint8_t *q)
{
    static const uint8_t base2[4] = {0x11, 0x22, 0x44, 0x88};
    const int nb = (l_seq + 1) / 2;
    for (int k = 0; k < nb; k += 32) {                /* 2 random bits per byte: both nibbles the same base code */
        uint64_t y = rnd(R);
        for (int j = 0; j < 32; j++, y >>= 2) sq[k + j] = base2[y & 3];
    }
    for (int k = 0; k < l_seq; k++) q[k] = (uint8_t)(20 + ((r >> (k & 31)) & 15));
}

static void fill_seq_real(rng_t *R, int l_seq, uint8_t *sq, uint8_t *q)
{
    static const uint8_t pair[16] = {0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88};
    const int nb = (l_seq + 1) / 2;
    for (int k = 0; k < nb; k += 16) {                /* 4 random bits per byte: two independent bases */
        uint64_t y = rnd(R);
        for (int j = 0; j < 16; j++, y >>= 4) sq[k + j] = pair[y & 15];
    }
    {                                                 /* an N now and then (0.2 % of the bases) */
        const uint64_t y = rnd(R);
        if ((y & 7) == 0 && nb > 0) sq[(y >> 8) % (uint64_t)nb] |= 0x0f;
    }
    if (content == C_HISEQ) {
        /* a level that starts at 37..41 and steps down along the read, per-base noise below it, sometimes a '#' tail */
        uint64_t y = rnd(R);
        int level = 37 + (int)(y % 5);
        const int tail = ((y >> 8) & 15) == 0 ? (int)((y >> 12) % (uint64_t)(l_seq / 2 + 1)) : 0;
        for (int k = 0; k < l_seq; k += 4) {
            y = rnd(R);
            for (int j = 0; j < 4 && k + j < l_seq; j++, y >>= 16) {
                const unsigned u = (unsigned)(y & 0xffff);
                if ((u & 15) == 0 && level > 24) level--;
                int e;
                const unsigned v = u >> 4;                           /* 12 bits */
                if (v < 2048) e = 0;
                else if (v < 3328) e = 1 + (int)(v & 3) % 3;
                else if (v < 3968) e = 4 + (int)(v % 9);
                else e = 13 + (int)(v % 20);
                int qq = level - e;
                if (qq < 2) qq = 2;
                q[k + j] = (uint8_t)qq;
            }
        }
        for (int k = l_seq - tail; k < l_seq; k++) q[k] = 2;
    } else {
        static const uint8_t bin[4] = {37, 23, 12, 2};
        int cur = 0;
        for (int k = 0; k < l_seq; k += 8) {
            uint64_t y = rnd(R);
            for (int j = 0; j < 8 && k + j < l_seq; j++, y >>= 8) {
                const unsigned u = (unsigned)(y & 0xff);
                if (u < 20) cur = u < 12 ? 0 : u < 17 ? 1 : u < 19 ? 2 : 3;        /* leaves its run 8 % of the time */
                q[k + j] = bin[cur];
            }
        }
    }
}

static int make_name(char *qn, rng_t *R, long long serial)
{
    if (content == C_LEGACY) {
        qn[0] = 'r';
        int ql = 1 + put_dec(qn + 1, serial);
        qn[ql++] = 0;
        return ql;
    }
    /* instrument_run:lane:tile:x:y — the tile and the coordinates of a read are unrelated to where it maps */
    const uint64_t y = rnd(R);
    int ql = 0;
    memcpy(qn, "HS25_09078:", 11);
    ql = 11;
    qn[ql++] = (char)('1' + (y & 7));
    qn[ql++] = ':';
    ql += put_dec(qn + ql, 1101 + (long long)((y >> 3) % 3) * 100 + (long long)((y >> 5) & 1) * 1000 + (long long)((y >> 6) % 16));
    qn[ql++] = ':';
    ql += put_dec(qn + ql, 1000 + (long long)((y >> 12) % 19000));
    qn[ql++] = ':';
    ql += put_dec(qn + ql, 2000 + (long long)((y >> 32) % 198000));
    qn[ql++] = 0;
    return ql;
}

/* one record appended to b; cig: n_cig words */
static void put_record(buf_t *b, int tid, int pos, int end, int mapq, int flag, const uint32_t *cig, int n_cig, int l_seq, int mtid, int mpos, int isize,
                       const char *qn, int ql, const uint8_t *sq, const uint8_t *q, const char *aux, int xl)
{
    const int nb = (l_seq + 1) / 2;
    const uint32_t block = 32 + (uint32_t)ql + 4u * (uint32_t)n_cig + (uint32_t)(nb + l_seq) + (uint32_t)xl;
    need(b, block + 4);
    uint32_t hdr[9] = {block,
                       (uint32_t)tid,
                       (uint32_t)pos,
                       ((uint32_t)reg2bin(pos, end > pos ? end : pos + 1) << 16) | ((uint32_t)mapq << 8) | (uint32_t)ql,
                       ((uint32_t)flag << 16) | (uint32_t)n_cig,
                       (uint32_t)l_seq,
                       (uint32_t)mtid,
                       (uint32_t)mpos,
                       (uint32_t)isize};
    memcpy(b->p + b->n, hdr, 36);
    b->n += 36;
    memcpy(b->p + b->n, qn, (size_t)ql);
    b->n += (size_t)ql;
    memcpy(b->p + b->n, cig, 4u * (size_t)n_cig);
    b->n += 4u * (size_t)n_cig;
    if (l_seq) {
        memcpy(b->p + b->n, sq, (size_t)nb);
        memcpy(b->p + b->n + nb, q, (size_t)l_seq);
        b->n += (size_t)(nb + l_seq);
    }
    if (xl) {
        memcpy(b->p + b->n, aux, (size_t)xl);
        b->n += (size_t)xl;
    }
}

/* the CIGAR of a read of rl query bases; returns the number of operations, *span = reference bases it covers */
static int make_cigar(uint64_t r, int rl, uint32_t *cg, int *span)
{
    if (cigar_mixed && (r >> 44) % 20 == 0 && rl >= 40) {
        const unsigned kind = (unsigned)((r >> 50) & 3), x = (unsigned)((r >> 52) & 0xfff);
        const int a = 10 + (int)(x % (unsigned)(rl - 20));                     /* 10 .. rl - 11 */
        switch (kind) {
        case 0: {
            const int s = 1 + (int)(x % 24);
            cg[0] = ((uint32_t)s << 4) | 4u;
            cg[1] = ((uint32_t)(rl - s) << 4) | 0u;
            *span = rl - s;
            return 2;
        }
        case 1: {
            const int d = 1 + (int)((x >> 5) % 8);
            cg[0] = ((uint32_t)a << 4) | 0u;
            cg[1] = ((uint32_t)d << 4) | 2u;
            cg[2] = ((uint32_t)(rl - a) << 4) | 0u;
            *span = rl + d;
            return 3;
        }
        case 2: {
            int i = 1 + (int)((x >> 5) % 6);
            if (i > rl - a - 1) i = 1;
            cg[0] = ((uint32_t)a << 4) | 0u;
            cg[1] = ((uint32_t)i << 4) | 1u;
            cg[2] = ((uint32_t)(rl - a - i) << 4) | 0u;
            *span = rl - i;
            return 3;
        }
        default: {
            const int n = 50 + (int)((r >> 20) % 5000);
            cg[0] = ((uint32_t)a << 4) | 0u;
            cg[1] = ((uint32_t)n << 4) | 3u;
            cg[2] = ((uint32_t)(rl - a) << 4) | 0u;
            *span = rl + n;
            return 3;
        }
        }
    }
    cg[0] = ((uint32_t)rl << 4) | 0u;
    *span = rl;
    return 1;
}

static int make_xa(char *xa, rng_t *R, int rl)
{
    int xl = 0;
    const uint64_t x = rnd(R);
    const int n_alt = 1 + (int)(x % 3);
    xa[xl++] = 'X', xa[xl++] = 'A', xa[xl++] = 'Z';
    for (int a = 0; a < n_alt; a++) {
        const uint64_t y = rnd(R);
        const int ac = (int)(y % (uint64_t)nc);
        const long long ap = 1 + (long long)((y >> 16) % (uint64_t)(sizes[ac] > rl ? sizes[ac] - rl : 1));
        xl += sprintf(xa + xl, "%s,%c", names[ac], (y >> 8) & 1 ? '-' : '+');
        xl += put_dec(xa + xl, ap);
        xl += sprintf(xa + xl, ",%dM,%d;", rl, (int)((y >> 9) & 3));
    }
    xa[xl++] = 0;
    xa[xl++] = 'N', xa[xl++] = 'M', xa[xl++] = 'C', xa[xl++] = (char)((x >> 8) & 3);
    return xl;
}

/* pileup=K: the reads of 1000 mean gaps either side of a locus land within 64 bp of it (monotone in pos: order is kept) */
static inline int pile(int c, int pos, double mean_gap)
{
    if (!n_pileup) return pos;
    long long kc = (long long)((double)n_pileup * (double)sizes[c] / (double)sizes[255]);       /* sizes[255]: the genome */
    if (kc < 1) return pos;
    const double pitch = (double)sizes[c] / (double)kc, w = 1000.0 * mean_gap;
    if (2.0 * w + 64.0 >= pitch) return pos;
    const long long l = (long long)((double)pos / pitch);
    const double centre = ((double)l + 0.5) * pitch;
    if ((double)pos < centre - w || (double)pos >= centre + w) return pos;
    return (int)(centre + ((double)pos - (centre - w)) * 64.0 / (2.0 * w));
}

typedef struct {
    int pos;
    uint32_t off, len;
} ridx_t;
static int by_pos(const void *a, const void *b)
{
    const ridx_t *x = a, *y = b;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return x->off < y->off ? -1 : x->off > y->off;
}

int main(int argc, char **argv)
{
    /* key=value options may stand anywhere; the others are the positional arguments */
    char *pv[16];
    int pn = 0;
    for (int i = 0; i < argc; i++) {
        const char *eq = strchr(argv[i], '=');
        if (i > 0 && eq) {
            const size_t kl = (size_t)(eq - argv[i]);
            if (kl == 7 && !strncmp(argv[i], "content", 7)) {
                if (!strcmp(eq + 1, "legacy")) content = C_LEGACY;
                else if (!strcmp(eq + 1, "hiseq")) content = C_HISEQ;
                else if (!strcmp(eq + 1, "novaseq")) content = C_NOVASEQ;
                else {
                    fprintf(stderr, "mkbam: unknown content '%s'\n", eq + 1);
                    return 1;
                }
            } else if (kl == 5 && !strncmp(argv[i], "cigar", 5)) {
                cigar_mixed = !strcmp(eq + 1, "mixed");
            } else if (kl == 6 && !strncmp(argv[i], "paired", 6)) {
                paired = atoi(eq + 1) != 0;
            } else if (kl == 6 && !strncmp(argv[i], "pileup", 6)) {
                n_pileup = atoi(eq + 1);
            } else {
                fprintf(stderr, "mkbam: unknown option '%s'\n", argv[i]);
                return 1;
            }
        } else if (pn < 16) {
            pv[pn++] = argv[i];
        }
    }
    if (pn < 4) {
        fprintf(stderr, "usage: mkbam <chrom.sizes> <n_reads> <out.bam> [seq_len=0] [seed=1] [xa_permille=0] [content=legacy|hiseq|novaseq] [cigar=simple|mixed] [paired=0|1] [pileup=K]\n");
        return 1;
    }
    const long long n_reads = atoll(pv[2]);
    const int seq_len = pn > 4 ? atoi(pv[4]) : 0;
    const uint64_t seed = pn > 5 ? strtoull(pv[5], 0, 0) : 1;
    const unsigned xa_pm = pn > 6 ? (unsigned)atoi(pv[6]) : 0;
    FILE *cf = fopen(pv[1], "r");
    if (!cf) return 2;
    while (nc < 255 && fscanf(cf, "%63s %lld", names[nc], &sizes[nc]) == 2) nc++;
    fclose(cf);
    if (nc == 0) return 2;
    long long genome = 0;
    for (int c = 0; c < nc; c++) genome += sizes[c];
    sizes[255] = genome;
    FILE *f = fopen(pv[3], "wb");
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
    int io_error = 0;
#pragma omp parallel
    {
        buf_t b = {0}, o = {0}, t = {0};
        uint8_t *sq = malloc((size_t)seq_len * 2 + 128);
        uint8_t *q = sq + (seq_len + 1) / 2 + 48;
        ridx_t *ri = paired ? malloc(sizeof *ri * (SEG + 2)) : NULL;
#pragma omp for ordered schedule(dynamic, 1)
        for (long long s = 0; s < n_seg; s++) {
            const seg_t *g = &seg[s];
            rng_t R = {seed * 0x2545f4914f6cdd1dull + (uint64_t)s * 0x9e3779b97f4a7c15ull + 1};
            const int c = g->chrom;
            const double mean_gap = (double)sizes[c] / (double)(nr_of[c] + 1);
            b.n = 0;
            o.n = 0;
            if (!paired) {
                for (long long i = 0; i < g->count; i++) {
                    const uint64_t r = rnd(&R);
                    /* read k of the chromosome sits in [k, k + 1) mean gaps: increasing in k without any running state */
                    int pos = (int)(((double)(g->first + i) + (double)(r >> 40) / (double)(1 << 24)) * mean_gap);
                    if (pos >= sizes[c]) pos = (int)sizes[c] - 1;
                    pos = pile(c, pos, mean_gap);
                    const int rl = (content != C_LEGACY && seq_len > 0) ? seq_len : 100 + (int)((r >> 8) % 51);
                    const int flag = (r & 1) ? 16 : 0;
                    char qn[64];
                    const int ql = make_name(qn, &R, g->serial + i);
                    char xa[256];
                    int xl = 0;
                    if (xa_pm && (unsigned)((r >> 20) % 1000) < xa_pm) xl = make_xa(xa, &R, rl);
                    uint32_t cg[4];
                    int span, n_cig = make_cigar(r, rl, cg, &span);
                    if (seq_len) {
                        if (content == C_LEGACY) fill_seq_legacy(&R, r, seq_len, sq, q);
                        else fill_seq_real(&R, seq_len, sq, q);
                    }
                    put_record(&b, c, pos, pos + span, mq[(r >> 4) & 7], flag, cg, n_cig, seq_len, -1, -1, 0, qn, ql, sq, q, xa, xl);
                }
            } else {
                /* fragments: records 2j and 2j + 1 of the chromosome are the two reads of fragment j, which starts in gap j of
                 * 2 mean gaps; a segment's reads stay below where the next segment's begin, so that sorting them here sorts the file */
                t.n = 0;
                const double gap2 = 2.0 * mean_gap;
                long long seg_end = (long long)((double)((g->first + g->count + 1) / 2) * gap2);
                if (seg_end > sizes[c]) seg_end = sizes[c];
                int nr = 0;
                for (long long i = 0; i < g->count; i += 2) {
                    const uint64_t r = rnd(&R), r2 = rnd(&R);
                    const int rl = (content != C_LEGACY && seq_len > 0) ? seq_len : 100 + (int)((r >> 8) % 51);
                    int start = (int)(((double)((g->first + i) / 2) + (double)(r >> 40) / (double)(1 << 24)) * gap2);
                    start = pile(c, start, mean_gap);
                    /* isize ~ N(350, 60): four uniforms on 0..104 (sd 30.3 each, 60.6 together) about their mean */
                    int isz = 350 + (int)((r2 & 0xffff) % 105) + (int)(((r2 >> 16) & 0xffff) % 105) + (int)(((r2 >> 32) & 0xffff) % 105) + (int)((r2 >> 48) % 105) - 208;
                    if (isz < rl) isz = rl;
                    if (isz > 700) isz = 700;
                    const int single = i + 1 >= g->count;                       /* an odd read at the end of a chromosome: unpaired */
                    if ((long long)start + isz >= seg_end) start = (int)(seg_end - 1 - isz);
                    if (start < 0) {
                        start = 0;
                        if ((long long)isz >= seg_end) isz = seg_end > rl ? (int)seg_end - 1 : rl;
                    }
                    const int right = start + isz - rl > start ? start + isz - rl : start;
                    const int first_is_left = (int)(r & 1);                      /* which read of the pair maps on the forward strand */
                    const int mate_unmapped = !single && (r >> 33) % 50 == 0;
                    char qn[64];
                    const int ql = make_name(qn, &R, g->serial + i);             /* both reads of a fragment carry one name */
                    for (int e = 0; e < (single ? 1 : 2); e++) {
                        const int left = e == 0;
                        const int is_read1 = left == first_is_left;
                        uint32_t cg[4];
                        int span, n_cig = make_cigar(e ? r2 : r, rl, cg, &span);
                        if (seq_len) {
                            if (content == C_LEGACY) fill_seq_legacy(&R, e ? r2 : r, seq_len, sq, q);
                            else fill_seq_real(&R, seq_len, sq, q);
                        }
                        char xa[256];
                        int xl = 0;
                        if (xa_pm && (unsigned)(((e ? r2 : r) >> 20) % 1000) < xa_pm) xl = make_xa(xa, &R, rl);
                        int flag, pos, mpos, isize, mapq = mq[((e ? r2 : r) >> 4) & 7], mtid = c;
                        if (single) {
                            flag = (r & 2) ? 16 : 0, pos = start, mpos = -1, isize = 0, mtid = -1;
                        } else if (mate_unmapped) {
                            /* the left read is mapped, its mate is not: both records at the mapped read's place */
                            pos = start, mpos = start, isize = 0;
                            if (left) flag = 0x1 | 0x8 | (is_read1 ? 0x40 : 0x80);
                            else flag = 0x1 | 0x4 | (is_read1 ? 0x40 : 0x80), mapq = 0, n_cig = 0, span = 0;
                        } else {
                            flag = 0x1 | 0x2 | (is_read1 ? 0x40 : 0x80) | (left ? 0x20 : 0x10);
                            pos = left ? start : right;
                            mpos = left ? right : start;
                            isize = left ? isz : -isz;
                        }
                        ri[nr].pos = pos;
                        ri[nr].off = (uint32_t)t.n;
                        put_record(&t, c, pos, pos + span, mapq, flag, cg, n_cig, seq_len, mtid, mpos, isize, qn, ql, sq, q, xa, xl);
                        ri[nr].len = (uint32_t)(t.n - ri[nr].off);
                        nr++;
                    }
                }
                qsort(ri, (size_t)nr, sizeof *ri, by_pos);
                need(&b, t.n);
                for (int k = 0; k < nr; k++) {
                    memcpy(b.p + b.n, t.p + ri[k].off, ri[k].len);
                    b.n += ri[k].len;
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
        free(t.p);
        free(sq);
        free(ri);
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
