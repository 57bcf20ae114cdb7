/* writers.c — the output files, byte for byte as the reference writes them: the report (generic.c:53-70), the
 * three stat tables and the two wig files (generic.c:72-113), the per-locus table (generic.c:1709-1746).
 * Row order is the iteration order of the reference's hashes (names_kent_order) and, inside a chromosome,
 * binKeeperNext's order: bins ascending, newest insertion first (cuskent/binRange.c:365-392). */
#define _GNU_SOURCE
#include "itx_host.h"

#include <errno.h>
#include <stdlib.h>
#include <string.h>

double cal_rpkm(unsigned long long reads_count, unsigned long long total_length, unsigned long long mapped_reads_num)
{
    return reads_count / (mapped_reads_num * 1e-9 * total_length);
}

double cal_rpm(unsigned long long reads_count, unsigned long long mapped_reads_num)
{
    return reads_count / (mapped_reads_num * 1e-6);
}

static FILE *must_open(const char *path, const char *mode)
{
    FILE *f = fopen(path, mode);
    if (!f) die("mustOpen: Can't open %s to write: %s", path, strerror(errno));      /* cuskent/common.c mustOpen */
    return f;
}

void write_report(const char *path, const uint64_t *cnt, unsigned mapQ, const char *subfam)
{
    FILE *f = must_open(path, "w");
    fprintf(f, "total reads (pair): %llu\n", (unsigned long long)cnt[0]);
    fprintf(f, "mappable reads (pair): %llu\n", (unsigned long long)cnt[6]);
    fprintf(f, "uniquely mapped reads (pair) (mapQ >= %u): %llu\n", mapQ, (unsigned long long)cnt[7]);
    fprintf(f, "non-redundant uniquely mapped reads (pair): %llu\n", (unsigned long long)cnt[11]);
    fprintf(f, "mapped reads (pair) overlap with repeats but discarded due to mapped to different subfamilies: %llu\n",
            (unsigned long long)cnt[12]);
    fprintf(f, "mapped reads (pair) overlap with [%s] repeats: %llu\n", subfam, (unsigned long long)cnt[9]);
    fprintf(f, "uniquely mapped reads (pair) overlap with [%s] repeats: %llu\n", subfam, (unsigned long long)cnt[10]);
    fclose(f);
}

/* "%u\n" */
static inline char *put_u32_line(char *p, uint32_t v)
{
    char t[10];
    int n = 0;
    do {
        t[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) *p++ = t[--n];
    *p++ = '\n';
    return p;
}

void write_wig_and_stat(const rmsk_t *rm, const itx_result *res, const uint64_t *cov_off, const char *f_stat, const char *f_wig,
                        const char *f_fam, const char *f_cla, const char *f_wig_uniq, unsigned long long reads_num,
                        unsigned long long reads_num_unique)
{
    const uint32_t S = rm->reps.n, F = rm->fams.n, C = rm->clas.n;
    uint32_t *order = xmalloc(sizeof(uint32_t) * (S + F + C + 1));
    FILE *f1 = must_open(f_stat, "w");
    FILE *f2 = f_wig ? must_open(f_wig, "w") : NULL;
    FILE *f5 = f_wig_uniq ? must_open(f_wig_uniq, "w") : NULL;
    fprintf(f1, "%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\n", "#subfamily", "family", "class", "consensus_length", "reads_count",
            "unique_reads_count", "total_length", "genome_count", "all_reads_RPKM", "all_reads_RPM", "unique_reads_RPKM",
            "unique_reads_RPM");
    names_kent_order(&rm->reps, 12, order);
    for (uint32_t k = 0; k < S; k++) {
        const uint32_t r = order[k];
        const unsigned long long rc = res->rep_cnt[r], ru = res->rep_cnt[S + r], tl = rm->rep_total[r];
        fprintf(f1, "%s\t%s\t%s\t%u\t%llu\t%llu\t%llu\t%llu\t%.3f\t%.3f\t%.3f\t%.3f\n", rm->reps.name[r], rm->fams.name[rm->rep_fam[r]],
                rm->clas.name[rm->rep_cla[r]], rm->rep_len[r], rc, ru, tl, (unsigned long long)rm->rep_genome[r],
                cal_rpkm(rc, tl, reads_num), cal_rpm(rc, reads_num), cal_rpkm(ru, tl, reads_num_unique), cal_rpm(ru, reads_num_unique));
    }
    /* the two wigs (generic.c:83-90), same order: one "%u\n" per base. The blocks are formatted in parallel, a group of
     * names at a time, and written in order. */
    if (f2 && f5) {
        enum { GROUP = 512 };
        char *buf[2][GROUP];
        size_t len[2][GROUP];
        for (uint32_t k0 = 0; k0 < S; k0 += GROUP) {
            const uint32_t k1 = k0 + GROUP < S ? k0 + GROUP : S;
#pragma omp parallel for schedule(dynamic, 4)
            for (long k = (long)k0; k < (long)k1; k++) {
                const uint32_t r = order[k];
                for (int w = 0; w < 2; w++) {
                    buf[w][k - k0] = NULL;
                    len[w][k - k0] = 0;
                    if (rm->rep_len[r] == 0) continue;
                    const uint32_t *v = (w ? res->cov_uniq : res->cov) + cov_off[r];
                    char *b = xmalloc(64 + strlen(rm->reps.name[r]) + 11 * (size_t)rm->rep_len[r]);
                    char *p = b + sprintf(b, "fixedStep chrom=%s start=1 step=1 span=1\n", rm->reps.name[r]);
                    for (uint32_t m = 0; m < rm->rep_len[r]; m++) p = put_u32_line(p, v[m]);
                    buf[w][k - k0] = b;
                    len[w][k - k0] = (size_t)(p - b);
                }
            }
            for (uint32_t k = k0; k < k1; k++) {
                if (buf[0][k - k0]) fwrite(buf[0][k - k0], 1, len[0][k - k0], f2);
                if (buf[1][k - k0]) fwrite(buf[1][k - k0], 1, len[1][k - k0], f5);
                free(buf[0][k - k0]);
                free(buf[1][k - k0]);
            }
        }
    }
    if (f2) fclose(f2);
    fclose(f1);
    if (f5) fclose(f5);
    FILE *f3 = must_open(f_fam, "w");
    fprintf(f3, "%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\n", "#family", "class", "reads_count", "unique_reads_count", "total_length",
            "genome_count", "all_reads_RPKM", "all_reads_RPM", "unique_reads_RPKM", "unique_reads_RPM");
    names_kent_order(&rm->fams, 12, order);
    for (uint32_t k = 0; k < F; k++) {
        const uint32_t r = order[k];
        const unsigned long long rc = res->fam_cnt[r], ru = res->fam_cnt[F + r], tl = rm->fam_total[r];
        fprintf(f3, "%s\t%s\t%llu\t%llu\t%llu\t%llu\t%.3f\t%.3f\t%.3f\t%.3f\n", rm->fams.name[r], rm->clas.name[rm->fam_cla[r]], rc, ru, tl,
                (unsigned long long)rm->fam_genome[r], cal_rpkm(rc, tl, reads_num), cal_rpm(rc, reads_num),
                cal_rpkm(ru, tl, reads_num_unique), cal_rpm(ru, reads_num_unique));
    }
    fclose(f3);
    FILE *f4 = must_open(f_cla, "w");
    fprintf(f4, "%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\n", "#class", "reads_count", "unique_reads_count", "total_length", "genome_count",
            "all_reads_RPKM", "all_reads_RPM", "unique_reads_RPKM", "unique_reads_RPM");
    names_kent_order(&rm->clas, 12, order);
    for (uint32_t k = 0; k < C; k++) {
        const uint32_t r = order[k];
        const unsigned long long rc = res->cla_cnt[r], ru = res->cla_cnt[C + r], tl = rm->cla_total[r];
        fprintf(f4, "%s\t%llu\t%llu\t%llu\t%llu\t%.3f\t%.3f\t%.3f\t%.3f\n", rm->clas.name[r], rc, ru, tl, (unsigned long long)rm->cla_genome[r],
                cal_rpkm(rc, tl, reads_num), cal_rpm(rc, reads_num), cal_rpkm(ru, tl, reads_num_unique), cal_rpm(ru, reads_num_unique));
    }
    fclose(f4);
    free(order);
}

/* cuskent/binRange.c:119-138 */
static int bin_of_range(int start, int end)
{
    static const int off[6] = {4096 + 512 + 64 + 8 + 1, 512 + 64 + 8 + 1, 64 + 8 + 1, 8 + 1, 1, 0};
    int sb = start >> 17, eb = (end - 1) >> 17;
    for (int i = 0; i < 6; ++i) {
        if (sb == eb) return off[i] + sb;
        sb >>= 3;
        eb >>= 3;
    }
    return -1;
}

struct lo {
    int bin;
    uint32_t row;
};
static int lo_cmp(const void *a, const void *b)
{
    const struct lo *x = a, *y = b;
    if (x->bin != y->bin) return x->bin < y->bin ? -1 : 1;
    return x->row > y->row ? -1 : (x->row < y->row ? 1 : 0);      /* list head = newest insertion (binRange.c:185) */
}

/* The loci of a table in binKeeper order: hashRmsk's chromosome order (names_kent_order), and for every chromosome its
 * rows with their bin numbers, to be sorted by lo_cmp (cuskent/binRange.c:365-392). */
static struct lo *loci_prepare(const rmsk_t *rm, uint32_t **corder_out, size_t **cnt_out, size_t **beg_out)
{
    const uint32_t NC = rm->chroms.n;
    uint32_t *corder = xmalloc(sizeof(uint32_t) * (NC + 1));
    names_kent_order(&rm->chroms, 12, corder);                  /* hashRmsk = newHash(0): 2^12 buckets */
    /* rows per chromosome */
    size_t *cnt = xcalloc(NC + 1, sizeof(size_t)), *beg = xcalloc(NC + 2, sizeof(size_t));
    for (size_t r = 0; r < rm->n_rows; r++) cnt[rm->row_chrom_name[r]]++;
    for (uint32_t c = 0; c < NC; c++) beg[c + 1] = beg[c] + cnt[c];
    struct lo *lo = xmalloc(sizeof *lo * (rm->n_rows ? rm->n_rows : 1));
    size_t *fill = xmalloc(sizeof(size_t) * (NC + 1));
    memcpy(fill, beg, sizeof(size_t) * (NC + 1));
    for (size_t r = 0; r < rm->n_rows; r++) {
        struct lo *e = &lo[fill[rm->row_chrom_name[r]]++];
        e->bin = bin_of_range((int)rm->rows[r].start, (int)rm->rows[r].end);
        e->row = (uint32_t)r;
    }
    free(fill);
    *corder_out = corder;
    *cnt_out = cnt;
    *beg_out = beg;
    return lo;
}

void write_filter_out(const rmsk_t *rm, const uint32_t *locus_cnt, char **locus_names, const char *path, int readlist, int threshold,
                      const char *subfam, unsigned long long reads_num)
{
    FILE *out = must_open(path, "w");
    int j = 0;
    if (readlist)
        fprintf(out, "%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\n", "#chr", "start", "end", "length", "repName", "repClass", "repFamily",
                "readsCount", "RPKM", "RPM", "readsList");
    else
        fprintf(out, "%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\n", "#chr", "start", "end", "length", "repName", "repClass", "repFamily",
                "readsCount", "RPKM", "RPM");
    const uint32_t NC = rm->chroms.n;
    uint32_t *corder;
    size_t *cnt, *beg;
    struct lo *lo = loci_prepare(rm, &corder, &cnt, &beg);
    for (uint32_t k = 0; k < NC; k++) {
        const uint32_t c = corder[k];
        qsort(lo + beg[c], cnt[c], sizeof *lo, lo_cmp);
        for (size_t i = beg[c]; i < beg[c] + cnt[c]; i++) {
            const uint32_t r = lo[i].row;
            const itx_row *os = &rm->rows[r];
            const int count = (int)locus_cnt[r];
            if (count < threshold) continue;
            j++;
            const unsigned length = os->end - os->start;
            /* filter by name: rows carry no name ids (the hashes stay empty): the strings come from the filter itself */
            fprintf(out, "%s\t%d\t%d\t%d\t%s\t%s\t%s\t%d\t%.3f\t%.3f", rm->chroms.name[c], (int)os->start, (int)os->end, (int)length,
                    rm->reps.name[os->rep], rm->clas.name[os->cla], rm->fams.name[os->fam], count,
                    cal_rpkm((unsigned long long)count, (unsigned long long)length, reads_num), cal_rpm((unsigned long long)count, reads_num));
            if (readlist) fprintf(out, "\t%s", locus_names && locus_names[r] ? locus_names[r] : "");
            fputc('\n', out);
        }
    }
    fclose(out);
    fprintf(stderr, "* Total %d [%s] TEs have at least %d reads mapped.\n", j, subfam, threshold);
    free(lo);
    free(cnt);
    free(beg);
    free(corder);
}

/* writeFilterOutMRE, generic.c:1748-1772 */
void write_cpg_loci(const rmsk_t *rm, const int *cpg_count, const double *cpg_total, const char *path, const char *subfam, double threshold)
{
    FILE *out = must_open(path, "w");
    int j = 0;
    fprintf(out, "%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\n", "#chr", "start", "end", "length", "repName", "repClass", "repFamily", "covered_CpG_site",
            "total_CpG_score");
    const uint32_t NC = rm->chroms.n;
    uint32_t *corder;
    size_t *cnt, *beg;
    struct lo *lo = loci_prepare(rm, &corder, &cnt, &beg);
    for (uint32_t k = 0; k < NC; k++) {
        const uint32_t c = corder[k];
        qsort(lo + beg[c], cnt[c], sizeof *lo, lo_cmp);
        for (size_t i = beg[c]; i < beg[c] + cnt[c]; i++) {
            const uint32_t r = lo[i].row;
            const itx_row *os = &rm->rows[r];
            if (!(cpg_total[r] > threshold)) continue;
            j++;
            fprintf(out, "%s\t%d\t%d\t%d\t%s\t%s\t%s\t%d\t%.3f\n", rm->chroms.name[c], (int)os->start, (int)os->end, (int)(os->end - os->start),
                    rm->reps.name[os->rep], rm->clas.name[os->cla], rm->fams.name[os->fam], cpg_count[r], cpg_total[r]);
        }
    }
    fclose(out);
    fprintf(stderr, "* Total %d [%s] TEs have CpG score larger than %.3f.\n", j, subfam, threshold);
    free(lo);
    free(cnt);
    free(beg);
    free(corder);
}
