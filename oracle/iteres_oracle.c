/* iteres_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE. See iteres_oracle.h.
 *
 * Plain-C, single-threaded restatement of the reference algorithm; deliberately
 * keeps the reference's data structure (the UCSC bin scheme with LIFO bin lists)
 * and its literal per-base loops so that it shares no design with the HIP path
 * it checks. Citations are to /root/reference.
 */
#include "iteres_oracle.h"

#include <stdlib.h>
#include <string.h>

/* cuskent/binRange.c:20-25 */
static const int BIN_OFFSETS_EXT[6] = {4096 + 512 + 64 + 8 + 1, 512 + 64 + 8 + 1, 64 + 8 + 1, 8 + 1, 1, 0};
#define BIN_FIRST_SHIFT 17
#define BIN_NEXT_SHIFT 3

struct row {
    uint32_t start, end, cons_start, cons_end;
    uint32_t rep, fam, cla;
    int32_t chrom;
    int64_t next; /* next element of the same bin list (older insertion), -1 = end */
};

struct chrom_bins {
    int64_t size;     /* chrom size from the size file; 0 = absent */
    int bin_count;    /* 0 until the first row arrives (binKeeperNew is lazy, generic.c:1623) */
    int64_t *head;    /* per-bin list head (most recent insertion first, slAddHead binRange.c:185) */
};

struct orc_table {
    int n_chrom, n_rep, n_fam, n_cla;
    struct chrom_bins *chr;
    uint32_t *rep_len;
    struct row *rows;
    int64_t n_rows, cap_rows;
};

/* cuskent/binRange.c:119-138 binFromRangeBinKeeperExtended; -1 where the reference errAborts. */
static int bin_from_range(int start, int end)
{
    int startBin = start, endBin = end - 1, i;
    startBin >>= BIN_FIRST_SHIFT;
    endBin >>= BIN_FIRST_SHIFT;
    for (i = 0; i < 6; ++i) {
        if (startBin == endBin)
            return BIN_OFFSETS_EXT[i] + startBin;
        startBin >>= BIN_NEXT_SHIFT;
        endBin >>= BIN_NEXT_SHIFT;
    }
    return -1;
}

orc_table *orc_table_new(int n_chrom, const int64_t *chrom_size, int n_rep, const uint32_t *rep_len,
                         int n_fam, int n_cla)
{
    orc_table *t = calloc(1, sizeof *t);
    t->n_chrom = n_chrom;
    t->n_rep = n_rep;
    t->n_fam = n_fam;
    t->n_cla = n_cla;
    t->chr = calloc(n_chrom > 0 ? n_chrom : 1, sizeof *t->chr);
    for (int c = 0; c < n_chrom; c++)
        t->chr[c].size = chrom_size[c];
    t->rep_len = calloc(n_rep > 0 ? n_rep : 1, sizeof *t->rep_len);
    if (n_rep > 0)
        memcpy(t->rep_len, rep_len, (size_t)n_rep * sizeof *rep_len);
    return t;
}

void orc_table_free(orc_table *t)
{
    if (!t)
        return;
    for (int c = 0; c < t->n_chrom; c++)
        free(t->chr[c].head);
    free(t->chr);
    free(t->rep_len);
    free(t->rows);
    free(t);
}

int64_t orc_table_add(orc_table *t, int chrom, uint32_t start, uint32_t end, uint32_t cons_start,
                      uint32_t cons_end, uint32_t rep, uint32_t fam, uint32_t cla)
{
    /* generic.c:1618-1622: chromosome not in the size file (hashIntValDefault(...,0) == 0) -> row freed */
    if (chrom < 0 || chrom >= t->n_chrom || (int)t->chr[chrom].size == 0)
        return -1;
    struct chrom_bins *cb = &t->chr[chrom];
    int maxPos = (int)cb->size;
    if (cb->bin_count == 0) {
        /* binRange.c:140-155 binKeeperNew(0, size) */
        if (maxPos < 0)
            return -2;
        int bc = bin_from_range(maxPos - 1, maxPos);
        if (bc < 0)
            return -2;
        cb->bin_count = bc + 1;
        cb->head = malloc((size_t)cb->bin_count * sizeof *cb->head);
        for (int b = 0; b < cb->bin_count; b++)
            cb->head[b] = -1;
    }
    /* binRange.c:171-186 binKeeperAdd(bk, s->start, s->end, s): unsigned -> int conversion at the call */
    int s = (int)start, e = (int)end;
    if (s < 0 || e > maxPos || s > e)
        return -2;
    int bin = bin_from_range(s, e);
    if (bin < 0 || bin >= cb->bin_count)
        return -2;
    if (t->n_rows == t->cap_rows) {
        t->cap_rows = t->cap_rows ? t->cap_rows * 2 : 1024;
        t->rows = realloc(t->rows, (size_t)t->cap_rows * sizeof *t->rows);
    }
    struct row *r = &t->rows[t->n_rows];
    r->start = start;
    r->end = end;
    r->cons_start = cons_start;
    r->cons_end = cons_end;
    r->rep = rep;
    r->fam = fam;
    r->cla = cla;
    r->chrom = chrom;
    r->next = cb->head[bin];
    cb->head[bin] = t->n_rows;
    return t->n_rows++;
}

void orc_table_add_many(orc_table *t, size_t n, const int32_t *chrom, const uint32_t *start, const uint32_t *end,
                        const uint32_t *cons_start, const uint32_t *cons_end, const uint32_t *rep,
                        const uint32_t *fam, const uint32_t *cla, int64_t *out)
{
    for (size_t i = 0; i < n; i++)
        out[i] = orc_table_add(t, chrom[i], start[i], end[i], cons_start[i], cons_end[i], rep[i], fam[i], cla[i]);
}

/* cuskent/common.c:2824-2831 */
static int range_intersection(int start1, int end1, int start2, int end2)
{
    int s = start1 > start2 ? start1 : start2;
    int e = end1 < end2 ? end1 : end2;
    return e - s;
}

/* cuskent/binRange.c:196-227. The reference prepends every match to the result list, so the
 * returned order is the REVERSE of the traversal order; we collect in traversal order and
 * reverse at the end. */
static int64_t find_hits(const orc_table *t, int chrom, int start, int end, int64_t **buf, int64_t *cap)
{
    const struct chrom_bins *cb = &t->chr[chrom];
    if (cb->bin_count == 0)
        return 0; /* no binKeeper for this chrom: hashLookup(hashRmsk, chr) == NULL, generic.c:945-946 */
    int minPos = 0, maxPos = (int)cb->size;
    if (start < minPos) start = minPos;
    if (end > maxPos) end = maxPos;
    if (start >= end) return 0;
    int startBin = start >> BIN_FIRST_SHIFT, endBin = (end - 1) >> BIN_FIRST_SHIFT;
    int64_t n = 0;
    for (int i = 0; i < 6; ++i) {
        int offset = BIN_OFFSETS_EXT[i];
        for (int j = startBin + offset; j <= endBin + offset; ++j) {
            for (int64_t el = cb->head[j]; el != -1; el = t->rows[el].next) {
                if (range_intersection((int)t->rows[el].start, (int)t->rows[el].end, start, end) > 0) {
                    if (n == *cap) {
                        *cap = *cap ? *cap * 2 : 64;
                        *buf = realloc(*buf, (size_t)*cap * sizeof **buf);
                    }
                    (*buf)[n++] = el;
                }
            }
        }
        startBin >>= BIN_NEXT_SHIFT;
        endBin >>= BIN_NEXT_SHIFT;
    }
    for (int64_t a = 0, b = n - 1; a < b; a++, b--) {
        int64_t tmp = (*buf)[a];
        (*buf)[a] = (*buf)[b];
        (*buf)[b] = tmp;
    }
    return n;
}

int64_t orc_find(const orc_table *t, int chrom, int start, int end, int64_t *rows, int64_t cap)
{
    if (chrom < 0 || chrom >= t->n_chrom)
        return 0;
    int64_t *buf = NULL, bcap = 0;
    int64_t n = find_hits(t, chrom, start, end, &buf, &bcap);
    for (int64_t i = 0; i < n && i < cap; i++)
        rows[i] = buf[i];
    free(buf);
    return n;
}

void orc_cov_offsets(const orc_table *t, uint64_t *off)
{
    uint64_t acc = 0;
    for (int r = 0; r < t->n_rep; r++) {
        off[r] = acc;
        acc += t->rep_len[r];
    }
    off[t->n_rep] = acc;
}

/* generic.c:296-301 getCov + cuskent/common.c:2833-2841 positiveRangeIntersection */
static float get_cov(unsigned int aStart, unsigned int aEnd, unsigned int start, unsigned int end)
{
    int ov = range_intersection((int)aStart, (int)aEnd, (int)start, (int)end);
    if (ov < 0)
        ov = 0;
    float overlap = ov;
    float denominator = (float)(aEnd - aStart);
    float cov = (denominator == 0) ? 0.0 : overlap / denominator;
    return cov;
}

#define FPAIRED 1
#define FUNMAP 4
#define FMUNMAP 8
#define FREVERSE 16
#define FREAD1 64

#define UMIN(a, b) ((a) < (b) ? (a) : (b)) /* kent's min macro, cuskent/common.h:1205 */

int orc_run(const orc_table *t, const orc_params *p, int n_tid, const int32_t *tid2chrom, size_t n,
            const int32_t *tid, const int32_t *pos, const int32_t *tmpend_a, const uint8_t *mapq,
            const uint16_t *flag, const int32_t *mpos, const int32_t *isize, const uint8_t *skip, int64_t *hit_row,
            uint64_t *cnt, uint64_t *rep_cnt, uint64_t *fam_cnt, uint64_t *cla_cnt, uint32_t *cov,
            uint32_t *cov_uniq, uint32_t *locus_cnt)
{
    const unsigned int mapQ = p->mapq_min, extension = p->extension, iSize = p->isize_max;
    const int treat = p->treat_pe_as_se, discardWrongEnd = p->discard_half_mapped;
    const float minCoverage = p->min_cov;
    uint64_t *cov_off = malloc(((size_t)t->n_rep + 1) * sizeof *cov_off);
    orc_cov_offsets(t, cov_off);
    int64_t *hits = NULL, hcap = 0;

    for (size_t r = 0; r < n; r++) {
        const unsigned fl = flag[r];
        const unsigned qual = mapq[r];
        unsigned int start, end, cend, rstart, rend;
        if (hit_row)
            hit_row[r] = -1;
        /* generic.c:748-759 read-end counters */
        int is_end1 = !(fl & FPAIRED) || (fl & FREAD1) || treat;
        if (is_end1) cnt[0]++; else cnt[1]++;
        /* generic.c:764 */
        if (fl & FUNMAP)
            continue;
        /* generic.c:768-779 */
        if (is_end1) cnt[2]++; else cnt[3]++;
        /* generic.c:781-801: chromosome name (with -C renaming) -> size; unknown -> skipped.
         * A tid outside the header crashes the reference; treated as unknown. */
        int chrom = (tid[r] >= 0 && tid[r] < n_tid) ? tid2chrom[tid[r]] : -1;
        if (chrom < 0)
            continue;
        /* generic.c:796-797: cend = (unsigned)(size - 1); a size of exactly 2 reads as "missing" */
        cend = (unsigned int)((int)t->chr[chrom].size - 1);
        if (cend == 1)
            continue;
        /* generic.c:802-813 */
        if (is_end1) cnt[4]++; else cnt[5]++;
        /* generic.c:815-905 */
        int se_style;
        if (treat) {
            se_style = 1;
        } else if (fl & FPAIRED) {
            if (!(fl & FMUNMAP)) {
                if (fl & FREAD1) {
                    if ((unsigned int)abs(isize[r]) > iSize || isize[r] == 0)
                        continue;
                    se_style = 0;
                } else {
                    continue;
                }
            } else {
                if (discardWrongEnd)
                    continue;
                se_style = 1;
            }
        } else {
            se_style = 1;
        }
        cnt[6]++;
        if (qual >= mapQ)
            cnt[7]++;
        if (se_style) {
            /* generic.c:819-833 (== 868-882 == 889-903) */
            start = (unsigned int)pos[r];
            int tmpend = tmpend_a[r];
            end = UMIN(cend, (unsigned int)tmpend);
            char strand = (fl & FREVERSE) ? '-' : '+';
            if (extension) {
                if (strand == '+') {
                    end = UMIN(start + extension, cend);
                } else {
                    if (end < extension)
                        start = 0;
                    else
                        start = end - extension;
                }
            }
        } else {
            /* generic.c:845-855 */
            if (isize[r] > 0) {
                start = (unsigned int)pos[r];
                int tmpend = (int)(start + (unsigned int)isize[r]);
                end = UMIN(cend, (unsigned int)tmpend);
            } else {
                start = (unsigned int)mpos[r];
                int tmpend = (int)(start - (unsigned int)isize[r]);
                end = UMIN(cend, (unsigned int)tmpend);
            }
        }
        /* generic.c:907-919 (-R) is order-dependent host logic and is not part of this restatement. */
        /* generic.c:921-922 */
        if (qual >= mapQ)
            cnt[11]++;
        if (skip && skip[r])
            continue;

        /* generic.c:939-970 */
        unsigned int qlen = end - start;
        int64_t nh = find_hits(t, chrom, (int)start, (int)end, &hits, &hcap);
        if (nh == 0)
            continue;
        int index = 0, tindex = 0;
        float coverage = 0.0, tcoverage = 0.0;
        for (int64_t k = 0; k < nh; k++) {
            index++;
            const struct row *sss = &t->rows[hits[k]];
            float c = get_cov(start, end, sss->start, sss->end);
            if (c > coverage) {
                tindex = index;
                tcoverage = c;
            }
            coverage = c;
        }
        if (tcoverage < minCoverage)
            continue;
        /* tindex == 0 would leave ss NULL in the reference (a crash); cannot happen for a hit
         * with positive overlap, guard anyway. */
        if (tindex == 0)
            continue;
        int64_t ridx = hits[tindex - 1];
        const struct row *ss = &t->rows[ridx];
        /* generic.c:972-982 XA/NM veto: host-side string logic, not restated here (no XA tags in
         * any fixture that goes through this function). */
        if (hit_row)
            hit_row[r] = ridx;
        if (p->filter_mode == 0) {
            /* generic.c:984-1024 */
            rep_cnt[ss->rep]++;
            if (qual >= mapQ)
                rep_cnt[t->n_rep + ss->rep]++;
            unsigned int length = t->rep_len[ss->rep];
            if (length != 0) {
                int i, j;
                rstart = start - ss->start;
                rend = rstart + qlen;
                rend = (rend < ss->end) ? rend : ss->end;
                for (i = (int)rstart; (unsigned int)i < rend; i++) {
                    j = (int)((unsigned int)i + ss->cons_start);
                    if ((unsigned int)j >= ss->cons_end)
                        break;
                    if ((unsigned int)j >= length)
                        break;
                    cov[cov_off[ss->rep] + (unsigned int)j]++;
                    if (qual >= mapQ)
                        cov_uniq[cov_off[ss->rep] + (unsigned int)j]++;
                }
            }
            fam_cnt[ss->fam]++;
            if (qual >= mapQ)
                fam_cnt[t->n_fam + ss->fam]++;
            cla_cnt[ss->cla]++;
            if (qual >= mapQ)
                cla_cnt[t->n_cla + ss->cla]++;
        } else {
            /* generic.c:662-666: slNameAddHead(&ss->sl, qname); count = slCount (generic.c:1725) */
            locus_cnt[ridx]++;
        }
        /* generic.c:1030-1032 */
        cnt[9]++;
        if (qual >= mapQ)
            cnt[10]++;
    }
    free(hits);
    free(cov_off);
    return 0;
}

uint32_t orc_hash_string(const char *s)
{
    uint32_t result = 0;
    int c;
    while ((c = *s++) != '\0')
        result += (result << 3) + (uint32_t)c;
    return result;
}
