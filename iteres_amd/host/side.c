/* side.c — what the reference's record loop decides with strings or in file order, kept on the host and handed to
 * the engine as one flag bit per record (ITX_F5_NOLOOKUP) or written straight to a file:
 *   - the coordinates of a record as the loop derives them (generic.c:764-905), needed by everything below,
 *   - -R, the duplicate filter on "chr:start:end:strand" keys (generic.c:907-919),
 *   - -B / -V, the bed lines of mapped reads (generic.c:925-936),
 *   - the XA/NM multi-mapping veto (generic.c:303-341, 972-982). */
#define _GNU_SOURCE
#include "itx_host.h"

#include <ctype.h>
#include <stdlib.h>
#include <string.h>

/* ---- generic.c:764-905 for one record: does it reach reads_mapped++, and with which interval and strand -------- */
int host_derive(const run_opts *o, int32_t chrom, int64_t chrom_size, unsigned flag5, int32_t pos, int32_t tmpend, int32_t mpos, int32_t isize,
                host_iv *d)
{
    enum { PAIRED = 1, UNMAP = 2, MUNMAP = 4, REVERSE = 8, READ1 = 16 };
    if (flag5 & UNMAP) return 0;                                      /* generic.c:764 */
    if (chrom < 0) return 0;                                          /* generic.c:781-801 (dropped by -C / not in the size file) */
    const uint32_t cend = (uint32_t)((int)chrom_size - 1);            /* generic.c:796 */
    if (cend == 1u) return 0;
    int se;
    if (o->treat || !(flag5 & PAIRED)) {
        se = 1;
    } else if (!(flag5 & MUNMAP)) {                                   /* generic.c:836-860 */
        if (!(flag5 & READ1)) return 0;
        const uint32_t a = isize < 0 ? 0u - (uint32_t)isize : (uint32_t)isize;
        if (a > o->isize || isize == 0) return 0;
        se = 0;
    } else {
        if (o->discard) return 0;                                     /* generic.c:862-863 */
        se = 1;
    }
    uint32_t start, end;
    char strand;
    if (se) {                                                         /* generic.c:819-833 */
        start = (uint32_t)pos;
        end = cend < (uint32_t)tmpend ? cend : (uint32_t)tmpend;
        strand = (flag5 & REVERSE) ? '-' : '+';
        if (o->extension) {
            if (strand == '+') {
                const uint32_t e2 = start + o->extension;
                end = e2 < cend ? e2 : cend;
            } else {
                start = end < o->extension ? 0u : end - o->extension;
            }
        }
    } else if (isize > 0) {                                           /* generic.c:845-855 */
        start = (uint32_t)pos;
        const uint32_t e2 = start + (uint32_t)isize;
        end = cend < e2 ? cend : e2;
        strand = '+';
    } else {
        start = (uint32_t)mpos;
        const uint32_t e2 = start - (uint32_t)isize;
        end = cend < e2 ? cend : e2;
        strand = '-';
    }
    d->start = start;
    d->end = end;
    d->strand = strand;
    return 1;
}

/* ---- -R: the set of keys seen so far ---------------------------------------------------------------------------
 * The reference's key is the string "chr:start:end:strand"; equal strings <=> equal (name, start, end, strand). */
typedef struct {
    uint32_t chr, start, end, strand;        /* strand 0 marks a free cell; the "no key yet" key is chr = UINT32_MAX */
} dup_key;
struct dup_set {
    dup_key *cell;
    size_t cap, n;
    dup_key cur;                              /* generic.c:909-912: only a MAPQ >= -Q record refreshes the key */
};

dup_set *dup_set_new(void)
{
    dup_set *s = xcalloc(1, sizeof *s);
    s->cap = 1u << 16;
    s->cell = xcalloc(s->cap, sizeof *s->cell);
    /* before the first refresh the reference looks up whatever its stack buffer holds; modelled as one key that no
     * record can produce (host side channels) */
    s->cur.chr = UINT32_MAX;
    s->cur.start = s->cur.end = 0;
    s->cur.strand = '?';
    return s;
}

void dup_set_free(dup_set *s)
{
    if (!s) return;
    free(s->cell);
    free(s);
}

static inline size_t dup_hash(const dup_key *k)
{
    uint64_t h = ((uint64_t)k->chr << 32 | k->start) * 0x9e3779b97f4a7c15ull;
    h ^= ((uint64_t)k->end << 8 | k->strand) * 0xc2b2ae3d27d4eb4full;
    h ^= h >> 29;
    return (size_t)h;
}

static int dup_insert(dup_set *s, const dup_key *k)        /* 1 when the key was already there */
{
    if ((s->n + 1) * 10 > s->cap * 7) {
        const size_t ncap = s->cap * 2;
        dup_key *nc = xcalloc(ncap, sizeof *nc);
        for (size_t i = 0; i < s->cap; i++)
            if (s->cell[i].strand) {
                size_t j = dup_hash(&s->cell[i]) & (ncap - 1);
                while (nc[j].strand) j = (j + 1) & (ncap - 1);
                nc[j] = s->cell[i];
            }
        free(s->cell);
        s->cell = nc;
        s->cap = ncap;
    }
    size_t j = dup_hash(k) & (s->cap - 1);
    while (s->cell[j].strand) {
        const dup_key *c = &s->cell[j];
        if (c->chr == k->chr && c->start == k->start && c->end == k->end && c->strand == k->strand) return 1;
        j = (j + 1) & (s->cap - 1);
    }
    s->cell[j] = *k;
    s->n++;
    return 0;
}

/* generic.c:907-919 for one record that reached this point: 1 = the record is dropped */
int dup_set_seen(dup_set *s, uint32_t chr_name_id, const host_iv *d, int uniq)
{
    if (uniq) {
        s->cur.chr = chr_name_id;
        s->cur.start = d->start;
        s->cur.end = d->end;
        s->cur.strand = (uint32_t)(unsigned char)d->strand;
    }
    return dup_insert(s, &s->cur);
}

/* ---- XA veto: rows of every chromosome that got a binKeeper, start-sorted, for "does anything here overlap" ------ */
struct xa_index {
    const rmsk_t *rm;
    uint32_t *off;             /* [chroms.n + 1] */
    int32_t *s, *e, *pmax;     /* per row in (chromosome, start) order; pmax = running max of e inside the chromosome */
    uint32_t *word;            /* case-insensitive identity of the row's repName (sameWord, cuskent/common.c:1316-1333) */
    uint32_t *rep_word;        /* [reps.n] the same identity per repName id */
};

typedef struct {
    int32_t s, e;
    uint32_t word;
} xa_row;
static int cmp_xa_row(const void *a, const void *b)
{
    const xa_row *x = a, *y = b;
    return x->s < y->s ? -1 : x->s > y->s;
}

/* rep_word[i] == rep_word[j] <=> sameWord(repName i, repName j) (cuskent/common.c:1316-1333: equal without case) */
uint32_t *xa_rep_words(const rmsk_t *rm)
{
    names_t words;
    names_init(&words);
    uint32_t *w = xcalloc((size_t)rm->reps.n + 1, sizeof(uint32_t));
    for (uint32_t i = 0; i < rm->reps.n; i++) {
        char *u = xstrdup(rm->reps.name[i]);
        for (char *p = u; *p; p++) *p = (char)toupper((unsigned char)*p);
        w[i] = names_intern(&words, u);
        free(u);
    }
    names_free(&words);
    return w;
}

xa_index *xa_index_new(const rmsk_t *rm)
{
    xa_index *x = xcalloc(1, sizeof *x);
    x->rm = rm;
    const uint32_t nc = rm->chroms.n;
    x->rep_word = xa_rep_words(rm);
    x->off = xcalloc((size_t)nc + 2, sizeof(uint32_t));
    for (size_t i = 0; i < rm->n_rows; i++) x->off[rm->row_chrom_name[i] + 1]++;
    for (uint32_t c = 0; c < nc; c++) x->off[c + 1] += x->off[c];
    xa_row *tmp = xcalloc(rm->n_rows + 1, sizeof *tmp);
    uint32_t *fill = xcalloc((size_t)nc + 1, sizeof(uint32_t));
    for (size_t i = 0; i < rm->n_rows; i++) {
        const uint32_t c = rm->row_chrom_name[i];
        xa_row *t = &tmp[x->off[c] + fill[c]++];
        t->s = (int32_t)rm->rows[i].start;
        t->e = (int32_t)rm->rows[i].end;
        t->word = x->rep_word[rm->rows[i].rep];
    }
    free(fill);
    x->s = xcalloc(rm->n_rows + 1, sizeof(int32_t));
    x->e = xcalloc(rm->n_rows + 1, sizeof(int32_t));
    x->pmax = xcalloc(rm->n_rows + 1, sizeof(int32_t));
    x->word = xcalloc(rm->n_rows + 1, sizeof(uint32_t));
    for (uint32_t c = 0; c < nc; c++) {
        qsort(tmp + x->off[c], x->off[c + 1] - x->off[c], sizeof *tmp, cmp_xa_row);
        int32_t pm = INT32_MIN;
        for (uint32_t k = x->off[c]; k < x->off[c + 1]; k++) {
            x->s[k] = tmp[k].s;
            x->e[k] = tmp[k].e;
            x->word[k] = tmp[k].word;
            if (tmp[k].e > pm) pm = tmp[k].e;
            x->pmax[k] = pm;
        }
    }
    free(tmp);
    return x;
}

void xa_index_free(xa_index *x)
{
    if (!x) return;
    free(x->off);
    free(x->s);
    free(x->e);
    free(x->pmax);
    free(x->word);
    free(x->rep_word);
    free(x);
}

/* binKeeperFind(bk, start, end) (cuskent/binRange.c:196-227) reduced to the question mapped2diffSubfam asks:
 * is there an overlapping row whose name is not `word`? */
static int xa_any_other(const xa_index *x, uint32_t c, int start, int end, uint32_t word)
{
    const int max_pos = (int)x->rm->chrom_size[c];
    if (start < 0) start = 0;
    if (end > max_pos) end = max_pos;
    if (start >= end) return 0;
    uint32_t lo = x->off[c], hi = x->off[c + 1];
    const uint32_t base = lo;
    while (lo < hi) {                                   /* first row starting at or after `end` */
        const uint32_t mid = lo + (hi - lo) / 2;
        if (x->s[mid] < end) lo = mid + 1; else hi = mid;
    }
    for (uint32_t k = lo; k > base;) {
        --k;
        if (x->pmax[k] <= start) break;                 /* nothing at or below k ends past start */
        const int ov = (x->e[k] < end ? x->e[k] : end) - (x->s[k] > start ? x->s[k] : start);
        if (ov > 0 && x->word[k] != word) return 1;
    }
    return 0;
}

/* generic.c:303-341 mapped2diffSubfam: 1 = some alternative hit with NM' <= nm falls on a different subfamily.
 * xa is consumed (chopped in place, like the reference's copy). */
int xa_veto(const xa_index *x, uint32_t chosen_rep, int nm, char *xa, int qlen)
{
    const uint32_t word = x->rep_word[chosen_rep];
    char *row[100];
    int nf = 0;
    if (*xa) {                                           /* chopByChar(ahstring, ';', row, 100), cuskent/common.c:2029-2053 */
        char *in = xa;
        for (nf = 0; nf < 100;) {
            row[nf++] = in;
            char *semi = strchr(in, ';');
            if (!semi) break;
            *semi = 0;
            in = semi + 1;
        }
    }
    for (int i = 0; i < nf; i++) {
        if (!*row[i]) continue;
        char *f[4];
        int n2 = 0;
        char *in = row[i];
        for (n2 = 0; n2 < 4;) {
            f[n2++] = in;
            char *comma = strchr(in, ',');
            if (!comma) break;
            *comma = 0;
            in = comma + 1;
        }
        if (n2 != 4) die("malformed XA alternative \"%s\": the reference asserts four comma-separated fields (generic.c:319)", row[i]);
        const int nm2 = (int)strtol(f[3], 0, 0);
        if (nm2 > nm) continue;
        const int start = abs((int)strtol(f[1], 0, 0));
        const int end = start + qlen;
        const int64_t c = names_find(&x->rm->chroms, f[0]);
        if (c < 0) continue;
        if (xa_any_other(x, (uint32_t)c, start, end, word)) return 1;
    }
    return 0;
}
