/* stream.c — the record loop of the reference (generic.c:700-1062 stat copy, 343-697 filter copy) as a host
 * pipeline around the GPU engine: decode a batch into one pinned staging slot while the other slot is in
 * flight (itx_engine_staging / submit_slot / wait_slot), keep what is order-dependent or text on the host
 * (progress banners, the "chromosome not in the size file" warnings, read names for filter -r). */
#define _GNU_SOURCE
#include "itx_host.h"

#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>

#define BATCH_RECORDS (4u << 20)

static void chk(int rc, const char *what)
{
    if (rc != ITX_OK) die("%s: %s", what, itx_last_error());
}

/* generic.c:781-791 (-C): NULL when the record is skipped ("GL*"), else the (possibly renamed) chromosome */
static const char *rename_chr(const char *name, int add_chr, char *buf, size_t bufsz)
{
    if (!add_chr) return name;
    if (strncmp(name, "GL", 2) == 0) return NULL;
    if (strcasecmp(name, "MT") == 0) return "chrM";
    if (strncmp(name, "chr", 3) != 0) {
        snprintf(buf, bufsz, "chr%s", name);
        return buf;
    }
    return name;
}

typedef struct {
    uint32_t row;
    char *name;
} hit_name;

void run_stream(const run_opts *o, const rmsk_t *rm, const sizes_t *chr_sizes, int filter_mode, int multi_file,
                unsigned progress_every, int want_qnames, itx_engine **eng_out, itx_table **tab_out, char ***locus_names)
{
    const int timing = getenv("ITX_TIMING") != NULL;
    struct timespec ts0, ts1;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    int ndev = itx_device_count();
    if (ndev <= 0) die("no usable MI355X (HIP) device: %s", ndev < 0 ? itx_last_error() : "none visible");
    /* table: every chromosome of the size file is known to the engine (a read may land on one without repeats) */
    const uint32_t n_chrom = chr_sizes->names.n;
    itx_table *tab = NULL;
    size_t bad = 0;
    int rc = itx_table_create(rm->rows, rm->n_rows, chr_sizes->value, (int)n_chrom, rm->rep_len, rm->reps.n, rm->fams.n, rm->clas.n, 0,
                              &tab, &bad);
    if (rc == ITX_E_RANGE) {
        const itx_row *r = &rm->rows[bad];
        die("(%d %d) out of range (%d %d) in binKeeperAdd", (int)r->start, (int)r->end, 0, (int)chr_sizes->value[r->chrom]);
    }
    chk(rc, "itx_table_create");
    itx_params p;
    memset(&p, 0, sizeof p);
    p.mapq_min = o->mapq;
    p.min_cov = o->min_cov;
    p.extension = o->extension;
    p.isize_max = o->isize;
    p.treat_pe_as_se = o->treat;
    p.discard_half_mapped = o->discard;
    p.mode = filter_mode ? ITX_MODE_FILTER : ITX_MODE_STAT;
    p.accum = ITX_ACCUM_DEFAULT;
    itx_engine *eng = NULL;
    chk(itx_engine_create(tab, &p, BATCH_RECORDS, &eng), "itx_engine_create");
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    if (timing) fprintf(stderr, "[itx timing] table build + engine %.3f s\n", (double)(ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts0.tv_nsec));
    itx_staging st[2];
    chk(itx_engine_staging(eng, 0, &st[0]), "itx_engine_staging");
    chk(itx_engine_staging(eng, 1, &st[1]), "itx_engine_staging");

    char **qn[2] = {NULL, NULL};
    size_t pend[2] = {0, 0};
    hit_name *hn = NULL;
    size_t n_hn = 0, cap_hn = 0;
    if (want_qnames) {
        qn[0] = xcalloc(BATCH_RECORDS, sizeof(char *));
        qn[1] = xcalloc(BATCH_RECORDS, sizeof(char *));
    }
    unsigned long long ends = 0;

    /* the list of files (stat: comma separated, generic.c:725; filter: one file) */
    char *arg = xstrdup(o->aln_arg);
    char *files[100];
    int n_files = 0;
    if (multi_file) {
        for (char *s = arg; n_files < 100;) {
            files[n_files++] = s;
            char *c = strchr(s, ',');
            if (!c) break;
            *c = 0;
            s = c + 1;
        }
    } else {
        files[n_files++] = arg;
    }
    names_t warned;
    names_init(&warned);
    for (int fi = 0; fi < n_files; fi++) {
        if (multi_file) fprintf(stderr, "\n* Processing %s\n", files[fi]);
        aln_reader *rd = aln_open(files[fi], o->is_sam);
        if (!rd) {
            fprintf(stderr, "Fail to open %s file %s\n", o->is_sam ? "SAM" : "BAM", o->aln_arg);
            die("Error\n");
        }
        const int nt = aln_n_targets(rd);
        int32_t *t2c = xmalloc(sizeof(int32_t) * (size_t)(nt + 1));
        char **t2name = xcalloc((size_t)nt + 1, sizeof(char *));
        for (int t = 0; t < nt; t++) {
            char buf[4096];
            const char *nm = rename_chr(aln_target_name(rd, t), o->add_chr, buf, sizeof buf);
            t2name[t] = nm ? xstrdup(nm) : NULL;
            if (!nm) {
                t2c[t] = -2;
            } else {
                const int64_t id = names_find(&chr_sizes->names, nm);
                /* generic.c:796-797: cend = size-1 with 2 as the "not found" default; a listed size of 2 reads the same */
                t2c[t] = (id >= 0 && (int)chr_sizes->value[id] != 2) ? (int32_t)id : -1;
            }
        }
        chk(itx_engine_set_tidmap(eng, t2c, nt > 0 ? nt : 0), "itx_engine_set_tidmap");
        if (nt == 0) {
            /* no references: nothing can map; still count the read ends */
            int32_t none = -1;
            chk(itx_engine_set_tidmap(eng, &none, 1), "itx_engine_set_tidmap");
        }
        int s = 0, any_paired = 0, aux_xa = 0;
        for (;;) {
            /* slot s: collect what its previous batch left behind, then refill */
            chk(itx_engine_wait_slot(eng, s), "itx_engine_wait_slot");
            if (want_qnames && pend[s]) {
                for (size_t i = 0; i < pend[s]; i++) {
                    const int32_t row = st[s].hit_row[i];
                    if (row >= 0) {
                        if (n_hn == cap_hn) {
                            cap_hn = cap_hn ? cap_hn * 2 : 1 << 16;
                            hn = xrealloc(hn, sizeof *hn * cap_hn);
                        }
                        hn[n_hn].row = (uint32_t)row;
                        hn[n_hn].name = qn[s][i];
                        n_hn++;
                    } else {
                        free(qn[s][i]);
                    }
                    qn[s][i] = NULL;
                }
                pend[s] = 0;
            }
            const size_t n = aln_read_batch(rd, &st[s], BATCH_RECORDS, want_qnames ? qn[s] : NULL, &any_paired, &aux_xa);
            if (n == 0) break;
            if (aux_xa && o->xa_veto)
                die("this alignment file carries XA tags: the XA/NM multi-mapping veto (generic.c:972-982) is not built into this "
                    "version — rerun with -x to count such reads as the reference does with -x");
            for (size_t i = 0; i < n; i++) {
                /* generic.c:760-761 progress; generic.c:793-801 one warning per unknown chromosome */
                if (++ends % progress_every == 0) fprintf(stderr, "\r* Processed read ends: %llu", ends);
                const int32_t t = st[s].tid[i];
                if (!(st[s].flag5[i] & 2) && t >= 0 && t < nt && t2c[t] == -1 && names_find(&warned, t2name[t]) < 0) {
                    names_intern(&warned, t2name[t]);
                    warnf("* Warning: read ends mapped to chromosome %s will be discarded as %s not existed in the chromosome size file",
                          t2name[t], t2name[t]);
                }
            }
            chk(itx_engine_submit_slot(eng, s, n, any_paired, want_qnames), "itx_engine_submit_slot");
            pend[s] = n;
            s ^= 1;
        }
        /* drain both slots before the tid map of the next file replaces this one */
        for (int k = 0; k < 2; k++) {
            chk(itx_engine_wait_slot(eng, k), "itx_engine_wait_slot");
            if (want_qnames && pend[k]) {
                for (size_t i = 0; i < pend[k]; i++) {
                    const int32_t row = st[k].hit_row[i];
                    if (row >= 0) {
                        if (n_hn == cap_hn) {
                            cap_hn = cap_hn ? cap_hn * 2 : 1 << 16;
                            hn = xrealloc(hn, sizeof *hn * cap_hn);
                        }
                        hn[n_hn].row = (uint32_t)row;
                        hn[n_hn].name = qn[k][i];
                        n_hn++;
                    } else {
                        free(qn[k][i]);
                    }
                    qn[k][i] = NULL;
                }
            }
            pend[k] = 0;
        }
        fprintf(stderr, "\r* Processed read ends: %llu\n", ends);
        for (int t = 0; t < nt; t++) free(t2name[t]);
        free(t2name);
        free(t2c);
        aln_close(rd);
    }
    names_free(&warned);
    free(arg);

    if (want_qnames && locus_names) {
        /* names per locus in BAM order (generic.c:1729 reverses the head-inserted list back to file order) */
        char **out = xcalloc(rm->n_rows ? rm->n_rows : 1, sizeof(char *));
        size_t *len = xcalloc(rm->n_rows ? rm->n_rows : 1, sizeof(size_t));
        for (size_t i = 0; i < n_hn; i++) len[hn[i].row] += strlen(hn[i].name) + 1;
        for (size_t r = 0; r < rm->n_rows; r++)
            if (len[r]) {
                out[r] = xmalloc(len[r] + 1);
                out[r][0] = 0;
                len[r] = 0;
            }
        for (size_t i = 0; i < n_hn; i++) {
            char *dst = out[hn[i].row] + len[hn[i].row];
            const size_t k = strlen(hn[i].name);
            if (len[hn[i].row]) *dst++ = ',', len[hn[i].row]++;
            memcpy(dst, hn[i].name, k + 1);
            len[hn[i].row] += k;
            free(hn[i].name);
        }
        free(len);
        *locus_names = out;
    }
    free(hn);
    free(qn[0]);
    free(qn[1]);
    *eng_out = eng;
    *tab_out = tab;
}
