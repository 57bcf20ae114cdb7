/* cmd_cpg.c — `iteres cpgstat` and `iteres cpgfilter` (cpgstat.c, cpgfilter.c, generic.c:115-152,1064-1139,1748-1772 of
 * the reference): CpG sites from a bedGraph file against the repeat table.
 *
 * The lookup — the FIRST row binKeeperFind returns for [start, end) — runs on the GPU (itx_engine_first_hit_slot: the
 * same table and window scan as the alignment path, no record filter, no best-hit rule). What the reference then adds
 * up are doubles, in file order (`cpgTotalScore += score`, `cpgScore[j] += score`): floating-point sums do not commute,
 * so they stay a sequential host pass over the chosen rows, which reproduces every "%.4f" / "%.3f" the reference prints. */
#define _GNU_SOURCE
#include "itx_host.h"

#include <ctype.h>
#include <errno.h>
#include <libgen.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#define BATCH_RECORDS (4u << 20)

static void chk(int rc, const char *what)
{
    if (rc != ITX_OK) die("%s: %s", what, itx_last_error());
}

/* ---- the bedGraph file (generic.c:1064-1078): lines chopped on white space, parsed in parallel pieces -------------- */
typedef struct {
    int32_t *tid;              /* index into rm->chroms (the chromosomes that have rows), -1 otherwise */
    int32_t *start, *end;      /* (unsigned int)strtol(...) as binKeeperFind's int arguments            */
    double *score;
    size_t n;
    char *err;
} cpg_piece;

static void cpg_parse_piece(char *text, size_t lo, size_t hi, size_t flen, const char *path, const rmsk_t *rm, cpg_piece *o)
{
    size_t cap = 0;
    const char *last_chr = NULL;
    int32_t last_tid = -1;
    size_t p = lo;
    while (p < hi) {
        char *line = text + p;
        const char *nl = memchr(line, '\n', flen - p);
        char *end = text + (nl ? (size_t)(nl - text) : flen);
        p = nl ? (size_t)(nl - text) + 1 : flen;
        /* lineFileNextReal (cuskent/linefile.c:826-839): blank lines and lines whose first non-blank is '#' are skipped */
        char *c = line;
        while (c < end && isspace((unsigned char)*c)) ++c;
        if (c >= end || *c == '#') continue;
        char *w[20];
        int n = 0;
        for (;;) {                                                   /* chopByWhite(line, row, 20) */
            if (n >= 20) break;
            while (c < end && isspace((unsigned char)*c)) ++c;
            if (c >= end || *c == 0) break;
            w[n++] = c;
            while (c < end && *c && !isspace((unsigned char)*c)) ++c;
            if (c >= end || *c == 0) break;
            *c++ = 0;
        }
        if (end < text + flen) *end = 0;
        if (n < 4) {
            if (asprintf(&o->err, "file %s doesn't appear to be in bedGraph format. At least 4 fields required, got %d", path, n) < 0) o->err = NULL;
            return;
        }
        if (o->n == cap) {
            cap = cap ? cap * 2 : 1 << 16;
            o->tid = xrealloc(o->tid, cap * sizeof *o->tid);
            o->start = xrealloc(o->start, cap * sizeof *o->start);
            o->end = xrealloc(o->end, cap * sizeof *o->end);
            o->score = xrealloc(o->score, cap * sizeof *o->score);
        }
        if (!last_chr || strcmp(last_chr, w[0]) != 0) {              /* hashLookup(hashRmsk, row[0]) */
            last_tid = (int32_t)names_find(&rm->chroms, w[0]);
            last_chr = w[0];
        }
        o->tid[o->n] = last_tid;
        o->start[o->n] = (int32_t)(unsigned int)strtol(w[1], NULL, 0);
        o->end[o->n] = (int32_t)(unsigned int)strtol(w[2], NULL, 0);
        o->score[o->n] = strtod(w[3], NULL);
        o->n++;
    }
}

typedef struct {
    int32_t *tid, *start, *end;
    double *score;
    size_t n;
} cpg_sites;

static void cpg_load(const char *path, const rmsk_t *rm, cpg_sites *out)
{
    size_t flen = 0;
    char *text = slurp_text(path, &flen);
    int T = omp_get_max_threads();
    if (T < 1) T = 1;
    if ((size_t)T > flen / 65536 + 1) T = (int)(flen / 65536 + 1);
    cpg_piece *parts = xcalloc((size_t)T, sizeof *parts);
    size_t *cut = xcalloc((size_t)T + 1, sizeof *cut);
    for (int t = 1; t < T; t++) {
        const size_t at = flen * (size_t)t / (size_t)T;
        const char *nl = memchr(text + at - 1, '\n', flen - (at - 1));
        cut[t] = nl ? (size_t)(nl - text) + 1 : flen;
        if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    }
    cut[T] = flen;
#pragma omp parallel for schedule(static, 1) num_threads(T)
    for (int t = 0; t < T; t++) cpg_parse_piece(text, cut[t], cut[t + 1], flen, path, rm, &parts[t]);
    memset(out, 0, sizeof *out);
    for (int t = 0; t < T; t++) {
        if (parts[t].err) die("%s", parts[t].err);                   /* the first one in file order */
        out->n += parts[t].n;
    }
    out->tid = xmalloc((out->n + 1) * sizeof *out->tid);
    out->start = xmalloc((out->n + 1) * sizeof *out->start);
    out->end = xmalloc((out->n + 1) * sizeof *out->end);
    out->score = xmalloc((out->n + 1) * sizeof *out->score);
    size_t at = 0;
    for (int t = 0; t < T; t++) {
        memcpy(out->tid + at, parts[t].tid, parts[t].n * sizeof *out->tid);
        memcpy(out->start + at, parts[t].start, parts[t].n * sizeof *out->start);
        memcpy(out->end + at, parts[t].end, parts[t].n * sizeof *out->end);
        memcpy(out->score + at, parts[t].score, parts[t].n * sizeof *out->score);
        at += parts[t].n;
        free(parts[t].tid);
        free(parts[t].start);
        free(parts[t].end);
        free(parts[t].score);
    }
    free(parts);
    free(cut);
    free(text);
}

/* ---- lookup of every site on the device: hit[i] = row (index into rm->rows) or -1 ----------------------------------- */
static int32_t *cpg_lookup(const rmsk_t *rm, const sizes_t *chr_sizes, const cpg_sites *s)
{
    int ndev = itx_device_count();
    if (ndev <= 0) die("no usable MI355X (HIP) device: %s", ndev < 0 ? itx_last_error() : "none visible");
    itx_table *tab = NULL;
    size_t bad = 0;
    int rc = itx_table_create(rm->rows, rm->n_rows, chr_sizes->value, (int)chr_sizes->names.n, rm->rep_len, rm->reps.n, rm->fams.n,
                              rm->clas.n, 0, &tab, &bad);
    if (rc == ITX_E_RANGE) {
        const itx_row *r = &rm->rows[bad];
        die("(%d %d) out of range (%d %d) in binKeeperAdd", (int)r->start, (int)r->end, 0, (int)chr_sizes->value[r->chrom]);
    }
    chk(rc, "itx_table_create");
    itx_params p;
    memset(&p, 0, sizeof p);
    p.mode = ITX_MODE_FILTER;                    /* no consensus accumulators are needed: only chosen rows come back */
    p.accum = ITX_ACCUM_ATOMIC;
    itx_engine *eng = NULL;
    chk(itx_engine_create(tab, &p, BATCH_RECORDS, &eng), "itx_engine_create");
    /* "tid" of a site = index into rm->chroms; the engine wants the index in the chrom-size file */
    int32_t *t2c = xmalloc(sizeof(int32_t) * (rm->chroms.n + 1));
    for (uint32_t i = 0; i < rm->chroms.n; i++) t2c[i] = (int32_t)names_find(&chr_sizes->names, rm->chroms.name[i]);
    if (rm->chroms.n == 0) t2c[0] = -1;
    chk(itx_engine_set_tidmap(eng, t2c, rm->chroms.n ? (int)rm->chroms.n : 1), "itx_engine_set_tidmap");
    itx_staging st[2];
    chk(itx_engine_staging(eng, 0, &st[0]), "itx_engine_staging");
    chk(itx_engine_staging(eng, 1, &st[1]), "itx_engine_staging");
    int32_t *hit = xmalloc(sizeof(int32_t) * (s->n + 1));
    size_t pend_off[2] = {0, 0}, pend_n[2] = {0, 0};
    size_t off = 0;
    for (int k = 0;; k ^= 1) {
        chk(itx_engine_wait_slot(eng, k), "itx_engine_wait_slot");
        if (pend_n[k]) memcpy(hit + pend_off[k], st[k].hit_row, pend_n[k] * sizeof(int32_t));
        pend_n[k] = 0;
        if (off >= s->n) {
            if (!pend_n[k ^ 1]) break;
            continue;
        }
        const size_t m = s->n - off < BATCH_RECORDS ? s->n - off : BATCH_RECORDS;
        memcpy(st[k].tid, s->tid + off, m * sizeof(int32_t));
        memcpy(st[k].pos, s->start + off, m * sizeof(int32_t));
        memcpy(st[k].tmpend, s->end + off, m * sizeof(int32_t));
        memset(st[k].mapq, 0, m);
        memset(st[k].flag5, 0, m);
        chk(itx_engine_first_hit_slot(eng, k, m), "itx_engine_first_hit_slot");
        pend_off[k] = off;
        pend_n[k] = m;
        off += m;
    }
    free(t2c);
    itx_engine_destroy(eng);
    itx_table_destroy(tab);
    return hit;
}

static FILE *must_open(const char *path)
{
    FILE *f = fopen(path, "w");
    if (!f) die("mustOpen: Can't open %s to write: %s", path, strerror(errno));
    return f;
}

static char *fmt_name(const char *prefix, const char *suffix)
{
    char *s = NULL;
    if (asprintf(&s, "%s%s", prefix, suffix) < 0) die("Mem Error.\n");
    return s;
}

/* ======================================================================================================== cpgstat */
static int cpgstat_usage(void)
{
    fprintf(stderr, "\n");
    fprintf(stderr, "obtain CpG statistics for each repeat subfamily, family and class.\n\n");
    fprintf(stderr, "Usage:   iteres cpgstat [options] <chromosome size file> <repeat size file> <rmsk.txt> <CpG bedGraph file>\n\n");
    fprintf(stderr, "Options: -w       keep the wiggle file [off]\n");
    fprintf(stderr, "         -o       output prefix [basename of input without extension]\n");
    fprintf(stderr, "         -h       help message\n");
    fprintf(stderr, "         -?       help message\n");
    fprintf(stderr, "\n");
    return 1;
}

int main_cpgstat(int argc, char **argv)
{
    int keep_wig = 0, c;
    char *optoutput = NULL;
    const time_t start_time = time(NULL);
    while ((c = getopt(argc, argv, "wo:h?")) >= 0) {
        switch (c) {
        case 'w': keep_wig = 1; break;
        case 'o': optoutput = strdup(optarg); break;
        case 'h':
        case '?': return cpgstat_usage();
        default: return 1;
        }
    }
    if (optind + 4 > argc) return cpgstat_usage();
    const char *chr_size_file = argv[optind], *rep_size_file = argv[optind + 1], *rmsk_file = argv[optind + 2];
    char *bedgraph_file = argv[optind + 3];
    char *output;
    if (optoutput) {
        output = optoutput;
    } else {
        char *tmp = xstrdup(bedgraph_file);
        output = filename_without_ext(basename(tmp));
        free(tmp);
    }
    char *outWig = fmt_name(output, ".CpGstat.wig"), *outBigWig = fmt_name(output, ".CpGstat.bigWig");
    char *outStat = fmt_name(output, ".CpG.subfamily.stat"), *outFam = fmt_name(output, ".CpG.family.stat");
    char *outCla = fmt_name(output, ".CpG.class.stat");

    gpu_warmup_start(0, NULL, 0, 0);
    sizes_t chr_sizes, rep_sizes;
    sizes_load(chr_size_file, &chr_sizes);
    sizes_load(rep_size_file, &rep_sizes);
    fprintf(stderr, "* Start to parse the rmsk file\n");
    rmsk_t rm;
    rmsk_load(rmsk_file, &chr_sizes, &rep_sizes, 0, "ALL", &rm);
    fprintf(stderr, "* Total %d repeats found.\n", rm.repeat_num);

    fprintf(stderr, "* Start to parse the bedGraph file\n");
    cpg_sites sites;
    cpg_load(bedgraph_file, &rm, &sites);
    int32_t *hit = cpg_lookup(&rm, &chr_sizes, &sites);

    /* generic.c:1089-1131 in file order: doubles */
    const uint32_t S = rm.reps.n, F = rm.fams.n, C = rm.clas.n;
    unsigned *rep_n = xcalloc(S + 1, sizeof *rep_n), *fam_n = xcalloc(F + 1, sizeof *fam_n), *cla_n = xcalloc(C + 1, sizeof *cla_n);
    double *rep_s = xcalloc(S + 1, sizeof *rep_s), *fam_s = xcalloc(F + 1, sizeof *fam_s), *cla_s = xcalloc(C + 1, sizeof *cla_s);
    uint64_t *off = xcalloc((size_t)S + 1, sizeof *off);
    for (uint32_t i = 0; i < S; i++) off[i + 1] = off[i] + rm.rep_len[i];
    double *cpg = xcalloc(off[S] + 1, sizeof *cpg);
    unsigned in_repeat = 0;
    for (size_t i = 0; i < sites.n; i++) {
        if (hit[i] < 0) continue;
        const itx_row *ss = &rm.rows[hit[i]];
        const double score = sites.score[i];
        rep_n[ss->rep]++;
        rep_s[ss->rep] += score;
        const uint32_t len = rm.rep_len[ss->rep];
        if (len != 0) {
            /* generic.c:1100-1113: the CpG's two bases, with the same unsigned arithmetic as the read coverage loop */
            const uint32_t rstart = (uint32_t)sites.start[i] - ss->start;
            uint32_t rend = rstart + 2;
            rend = rend < ss->end ? rend : ss->end;
            for (int k = (int)rstart; (uint32_t)k < rend; k++) {
                const int j = k + (int)ss->cons_start;
                if ((uint32_t)j >= ss->cons_end) break;
                if ((uint32_t)j >= len) break;
                cpg[off[ss->rep] + (uint32_t)j] += score;
            }
        }
        fam_n[ss->fam]++;
        fam_s[ss->fam] += score;
        cla_n[ss->cla]++;
        cla_s[ss->cla] += score;
        in_repeat++;
    }
    fprintf(stderr, "* Processed CpG sites: %u\n", (unsigned)sites.n);
    fprintf(stderr, "* CpG sites in Repeats: %u\n", in_repeat);

    fprintf(stderr, "* Writing stats and Wig file\n");
    {   /* MREwriteWigandStat, generic.c:115-152 */
        uint32_t *order = xmalloc(sizeof(uint32_t) * (S + F + C + 1));
        FILE *f1 = must_open(outStat), *f2 = must_open(outWig);
        fprintf(f1, "%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\n", "#subfamily", "family", "class", "consensus_length", "covered_CpG_sites",
                "CpG_total_score", "total_length", "genome_count");
        names_kent_order(&rm.reps, 12, order);
        for (uint32_t k = 0; k < S; k++) {
            const uint32_t r = order[k];
            fprintf(f1, "%s\t%s\t%s\t%u\t%u\t%.4f\t%llu\t%llu\n", rm.reps.name[r], rm.fams.name[rm.rep_fam[r]], rm.clas.name[rm.rep_cla[r]],
                    rm.rep_len[r], rep_n[r], rep_s[r], (unsigned long long)rm.rep_total[r], (unsigned long long)rm.rep_genome[r]);
            if (rm.rep_len[r] != 0) {
                fprintf(f2, "fixedStep chrom=%s start=1 step=1 span=1\n", rm.reps.name[r]);
                for (uint32_t m = 0; m < rm.rep_len[r]; m++) fprintf(f2, "%.4f\n", cpg[off[r] + m]);
            }
        }
        fclose(f2);
        fclose(f1);
        FILE *f3 = must_open(outFam);
        fprintf(f3, "%s\t%s\t%s\t%s\t%s\t%s\n", "#family", "class", "covered_CpG_sites", "CpG_total_score", "total_length", "genome_count");
        names_kent_order(&rm.fams, 12, order);
        for (uint32_t k = 0; k < F; k++) {
            const uint32_t r = order[k];
            fprintf(f3, "%s\t%s\t%u\t%.4f\t%llu\t%llu\n", rm.fams.name[r], rm.clas.name[rm.fam_cla[r]], fam_n[r], fam_s[r],
                    (unsigned long long)rm.fam_total[r], (unsigned long long)rm.fam_genome[r]);
        }
        fclose(f3);
        FILE *f4 = must_open(outCla);
        fprintf(f4, "%s\t%s\t%s\t%s\t%s\n", "#class", "covered_CpG_sites", "CpG_total_score", "total_length", "genome_count");
        names_kent_order(&rm.clas, 12, order);
        for (uint32_t k = 0; k < C; k++) {
            const uint32_t r = order[k];
            fprintf(f4, "%s\t%u\t%.4f\t%llu\t%llu\n", rm.clas.name[r], cla_n[r], cla_s[r], (unsigned long long)rm.cla_total[r],
                    (unsigned long long)rm.cla_genome[r]);
        }
        fclose(f4);
        free(order);
    }

    fprintf(stderr, "* Generating bigWig files\n");
    {   /* cpgstat.c:76: the converter reads every "%.4f" of the wig back as a double and stores a float */
        float *fv = xmalloc(sizeof(float) * (off[S] + 1));
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)off[S]; i++) {
            char buf[64];
            snprintf(buf, sizeof buf, "%.4f", cpg[i]);
            fv[i] = (float)strtod(buf, NULL);
        }
        const char **nm = xcalloc((size_t)S + 1, sizeof *nm);
        uint32_t *ln = xcalloc((size_t)S + 1, sizeof *ln);
        const float **va = xcalloc((size_t)S + 1, sizeof *va);
        size_t k = 0;
        for (uint32_t i = 0; i < S; i++)
            if (rm.rep_len[i]) {
                nm[k] = rm.reps.name[i];
                ln[k] = rm.rep_len[i];
                va[k] = fv + off[i];
                k++;
            }
        write_bigwig(outBigWig, outWig, nm, ln, va, k);
        free(nm); free(ln); free(va); free(fv);
    }
    if (!keep_wig) unlink(outWig);
    fprintf(stderr, "* Done, time used %.0f seconds.\n", difftime(time(NULL), start_time));
    return 0;
}

/* ====================================================================================================== cpgfilter */
static int cpgfilter_usage(void)
{
    fprintf(stderr, "\n");
    fprintf(stderr, "obtain CpG statistics for each repeat locus.\n\n");
    fprintf(stderr, "Usage:   iteres cpgfilter [options] <chromosome size file> <repeat size file> <rmsk.txt> <CpG bedGraph file>\n\n");
    fprintf(stderr, "Options: -n       use repName (subfamily) as filter [null]\n");
    fprintf(stderr, "         -f       use repFamily as filter [null]\n");
    fprintf(stderr, "         -c       use repClass as filter [null]\n");
    fprintf(stderr, "         -t       only output repeats have more than [0] CpG score\n");
    fprintf(stderr, "         -o       output prefix [basename of input without extension]\n");
    fprintf(stderr, "         -h       help message\n");
    fprintf(stderr, "         -?       help message\n");
    fprintf(stderr, "\n");
    return 1;
}

int main_cpgfilter(int argc, char **argv)
{
    int c, filter_field = 0;
    double threshold = 0;
    char *optoutput = NULL, *optname = NULL, *optclass = NULL, *optfamily = NULL;
    const time_t start_time = time(NULL);
    while ((c = getopt(argc, argv, "n:c:f:t:o:h?")) >= 0) {
        switch (c) {
        case 'n': optname = strdup(optarg); break;
        case 'c': optclass = strdup(optarg); break;
        case 'f': optfamily = strdup(optarg); break;
        case 't': threshold = strtod(optarg, NULL); break;
        case 'o': optoutput = strdup(optarg); break;
        case 'h':
        case '?': return cpgfilter_usage();
        default: return 1;
        }
    }
    if (optind + 4 > argc) return cpgfilter_usage();
    const char *chr_size_file = argv[optind], *rep_size_file = argv[optind + 1], *rmsk_file = argv[optind + 2];
    char *bedgraph_file = argv[optind + 3];
    if ((optname && optclass) || (optname && optfamily) || (optclass && optfamily))
        die("Please specify only one filter, either -n, -c or -f.");
    char *output;
    if (optoutput) {
        output = optoutput;
    } else {
        char *tmp = xstrdup(bedgraph_file);
        output = filename_without_ext(basename(tmp));
        free(tmp);
    }
    const char *subfam = "ALL";
    if (optname) {
        subfam = optname;
        filter_field = 10;
    } else if (optclass) {
        subfam = optclass;
        filter_field = 11;
    } else if (optfamily) {
        subfam = optfamily;
        filter_field = 12;
    }
    if (strcmp(subfam, "ALL") == 0) {
        fprintf(stderr, "* You didn't specify any filter, will output all repeats\n");
        filter_field = 0;
    }
    gpu_warmup_start(0, NULL, 0, 0);
    sizes_t chr_sizes, rep_sizes;
    sizes_load(chr_size_file, &chr_sizes);
    sizes_load(rep_size_file, &rep_sizes);
    fprintf(stderr, "* Start to parse the rmsk file\n");
    rmsk_t rm;
    rmsk_load(rmsk_file, &chr_sizes, &rep_sizes, filter_field, subfam, &rm);
    fprintf(stderr, "* Total %d repeats found.\n", rm.repeat_num);

    fprintf(stderr, "* Start to parse the bedGraph file\n");
    cpg_sites sites;
    cpg_load(bedgraph_file, &rm, &sites);
    int32_t *hit = cpg_lookup(&rm, &chr_sizes, &sites);
    int *cnt = xcalloc(rm.n_rows + 1, sizeof *cnt);
    double *tot = xcalloc(rm.n_rows + 1, sizeof *tot);
    unsigned in_repeat = 0;
    for (size_t i = 0; i < sites.n; i++) {                         /* generic.c:1090-1092, file order */
        if (hit[i] < 0) continue;
        cnt[hit[i]]++;
        tot[hit[i]] += sites.score[i];
        in_repeat++;
    }
    fprintf(stderr, "* Processed CpG sites: %u\n", (unsigned)sites.n);
    fprintf(stderr, "* CpG sites in Repeats: %u\n", in_repeat);

    fprintf(stderr, "* Preparing the output file\n");
    char *out = NULL;
    if (asprintf(&out, "%s_%s.CpG.loci", output, subfam) < 0) die("Preparing output wrong");
    write_cpg_loci(&rm, cnt, tot, out, subfam, threshold);
    fprintf(stderr, "* Done, time used %.0f seconds.\n", difftime(time(NULL), start_time));
    return 0;
}
