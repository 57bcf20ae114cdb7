/* bigwig.c — the bigWig files `iteres stat` leaves beside its stat files (stat.c:156-158), written straight from the
 * coverage vectors instead of re-parsing the text wig. The layout and every derived number follow what the
 * reference's converter produces for a wig of "fixedStep chrom=<repName> start=1 step=1 span=1" blocks
 * (bigWigFileCreate(wig, repSizes, blockSize 256, itemsPerSlot 1024, clip 0, compress 1)):
 *   sections        cuskent/bwgCreate.c:186-262 (<= itemsPerSlot values each), sorted by chromosome name (:138-151)
 *   chromosome ids  cuskent/bwgCreate.c:584-627, B+ tree cuskent/bPlusTree.c:419-576
 *   zoom levels     cuskent/bwgCreate.c:826-885 with the summaries of cuskent/bbiWrite.c:370-446
 *   section writer  cuskent/bwgCreate.c:45-135, zoom writer cuskent/bbiWrite.c:478-536
 *   R tree          cuskent/cirTree.c:141-367
 *   file skeleton   cuskent/bwgCreate.c:887-1019
 * Blocks are deflated with zlib's compress() like cuskent/zlibFace.c:37-50; with the same zlib the files come out
 * byte for byte the same, with another one they decode to the same content. */
#define _GNU_SOURCE
#include "itx_host.h"

#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define BW_SIG 0x888FFC26u
#define BPT_SIG 0x78CA8C91u
#define CIR_SIG 0x2468ACE0u
#define BW_VERSION 4

typedef struct {
    uint32_t chrom_id, start, end, item_count;
    const float *val;             /* item_count values, as a section stores them */
    uint64_t file_offset;
} bw_section;

typedef struct {
    uint32_t chrom_id, start, end, valid_count;
    float min_val, max_val, sum_data, sum_squares;
    uint64_t file_offset;
} bw_summary;

typedef struct {
    bw_summary *v;
    size_t n, cap;
} bw_sumlist;

typedef struct {
    const char *name;
    uint32_t id, size;
} bw_chrom;

static void put(FILE *f, const void *p, size_t n)
{
    if (fwrite(p, 1, n, f) != n) die("mustWrite: Couldn't write to bigWig file");
}
#define PUT(f, x) put(f, &(x), sizeof(x))
static void put_zero(FILE *f, size_t n)
{
    static const char z[64] = {0};
    while (n) {
        const size_t k = n < sizeof z ? n : sizeof z;
        put(f, z, k);
        n -= k;
    }
}

/* ---- cuskent/bbiWrite.c:370-421 bbiAddToSummary, the list kept as an array whose last element is the open summary */
static void add_to_summary(bw_sumlist *L, uint32_t chrom_id, uint32_t chrom_size, uint32_t start, uint32_t end, uint32_t valid_count,
                           double min_val, double max_val, double sum_data, double sum_squares, int reduction)
{
    if (end > chrom_size) end = chrom_size;
    while (start < end) {
        bw_summary *sum = L->n ? &L->v[L->n - 1] : NULL;
        if (!sum || sum->chrom_id != chrom_id || sum->end <= start) {
            if (L->n == L->cap) {
                L->cap = L->cap ? L->cap * 2 : 1024;
                L->v = xrealloc(L->v, L->cap * sizeof *L->v);
                sum = L->n ? &L->v[L->n - 1] : NULL;
            }
            bw_summary nw;
            memset(&nw, 0, sizeof nw);
            nw.chrom_id = chrom_id;
            if (!sum || sum->chrom_id != chrom_id || sum->end + (uint32_t)reduction <= start)
                nw.start = start;
            else
                nw.start = sum->end;
            nw.end = nw.start + (uint32_t)reduction;
            if (nw.end > chrom_size) nw.end = chrom_size;
            nw.min_val = (float)min_val;
            nw.max_val = (float)max_val;
            L->v[L->n++] = nw;
            sum = &L->v[L->n - 1];
        }
        /* rangeIntersection, cuskent/common.c:2824-2831 */
        const int s = (int)(start > sum->start ? start : sum->start), e = (int)(end < sum->end ? end : sum->end);
        const int overlap = e - s;
        if (overlap <= 0) die("internal error: bigWig summary item %u %u does not intersect %u %u", start, end, sum->start, sum->end);
        const int item_size = (int)(end - start);
        const double f = (double)overlap / item_size;
        sum->valid_count = (uint32_t)(sum->valid_count + f * valid_count);
        if (sum->min_val > min_val) sum->min_val = (float)min_val;
        if (sum->max_val < max_val) sum->max_val = (float)max_val;
        sum->sum_data = (float)(sum->sum_data + f * sum_data);
        sum->sum_squares = (float)(sum->sum_squares + f * sum_squares);
        start += (uint32_t)overlap;
    }
}

/* Summaries never span two chromosomes (a new one starts whenever the chromosome id changes, bbiWrite.c:381), so the
 * chromosomes are reduced independently, in parallel, and their lists joined in order: same numbers, same order. */
static void sumlist_join(bw_sumlist *out, bw_sumlist *parts, size_t n_parts)
{
    size_t total = 0;
    for (size_t i = 0; i < n_parts; i++) total += parts[i].n;
    out->v = xmalloc((total ? total : 1) * sizeof *out->v);
    out->n = out->cap = total;
    size_t at = 0;
    for (size_t i = 0; i < n_parts; i++) {
        if (parts[i].n) memcpy(out->v + at, parts[i].v, parts[i].n * sizeof *out->v);
        at += parts[i].n;
        free(parts[i].v);
    }
    free(parts);
}

/* cuskent/bwgCreate.c:751-795: every base of every section, as a one-base range (bbiAddRangeToSummary, bbiWrite.c:424-433).
 * sec_of[c] .. sec_of[c+1]: the sections of chromosome c. */
static bw_sumlist reduce_sections(const bw_section *sec, const size_t *sec_of, const bw_chrom *chroms, size_t n_chroms, int reduction)
{
    bw_sumlist *parts = xcalloc(n_chroms ? n_chroms : 1, sizeof *parts);
#pragma omp parallel for schedule(dynamic, 8)
    for (long c = 0; c < (long)n_chroms; c++) {
        bw_sumlist L = {NULL, 0, 0};
        for (size_t k = sec_of[c]; k < sec_of[c + 1]; k++) {
            const uint32_t chrom_size = chroms[sec[k].chrom_id].size;
            uint32_t start = sec[k].start;
            /* bbiAddToSummary for one-base items, the arithmetic of add_to_summary spelled out for overlap == item size == 1
             * (f = 1.0): same operations on the same float fields in the same order, without the call and its loop per base */
            const float *vals = sec[k].val;
            const uint32_t n_items = sec[k].item_count;
            uint32_t i = 0;
            while (i < n_items) {
                if (start >= chrom_size) break;                         /* add_to_summary clips end to the chromosome: nothing is added */
                bw_summary *sum = L.n ? &L.v[L.n - 1] : NULL;
                if (!sum || sum->chrom_id != sec[k].chrom_id || sum->end <= start) {
                    const double v0 = (double)vals[i];
                    add_to_summary(&L, sec[k].chrom_id, chrom_size, start, start + 1, 1u, v0, v0, v0, v0 * v0, reduction);   /* opens the summary */
                    start += 1;
                    i++;
                    continue;
                }
                /* the open summary takes every base up to its end */
                uint32_t upto = sum->end - start;
                if (upto > n_items - i) upto = n_items - i;
                uint32_t vc = sum->valid_count;
                float mn = sum->min_val, mx = sum->max_val, sd = sum->sum_data, sq = sum->sum_squares;
                for (uint32_t j = 0; j < upto; j++) {
                    const double val = (double)vals[i + j];
                    vc = (uint32_t)((double)vc + 1.0);
                    if (mn > val) mn = (float)val;
                    if (mx < val) mx = (float)val;
                    sd = (float)((double)sd + val);
                    sq = (float)((double)sq + val * val);
                }
                sum->valid_count = vc;
                sum->min_val = mn;
                sum->max_val = mx;
                sum->sum_data = sd;
                sum->sum_squares = sq;
                start += upto;
                i += upto;
            }
        }
        parts[c] = L;
    }
    bw_sumlist out;
    sumlist_join(&out, parts, n_chroms);
    return out;
}

/* cuskent/bbiWrite.c:435-446 */
static bw_sumlist reduce_summaries(const bw_sumlist *in, const bw_chrom *chroms, size_t n_chroms, int reduction)
{
    /* first summary of every chromosome in the input list (ids ascend) */
    size_t *first = xcalloc(n_chroms + 2, sizeof *first);
    {
        size_t c = 0;
        for (size_t i = 0; i < in->n; i++)
            while (c <= in->v[i].chrom_id) first[c++] = i;
        while (c <= n_chroms) first[c++] = in->n;
    }
    bw_sumlist *parts = xcalloc(n_chroms ? n_chroms : 1, sizeof *parts);
#pragma omp parallel for schedule(dynamic, 8)
    for (long c = 0; c < (long)n_chroms; c++) {
        bw_sumlist L = {NULL, 0, 0};
        for (size_t i = first[c]; i < first[c + 1]; i++) {
            const bw_summary *s = &in->v[i];
            add_to_summary(&L, s->chrom_id, chroms[s->chrom_id].size, s->start, s->end, s->valid_count, s->min_val, s->max_val, s->sum_data,
                           s->sum_squares, reduction);
        }
        parts[c] = L;
    }
    free(first);
    bw_sumlist out;
    sumlist_join(&out, parts, n_chroms);
    return out;
}

/* ---- cuskent/zlibFace.c:37-56 */
static size_t z_buf_size(size_t n) { return (size_t)(1.001 * (double)n + 13); }
/* zlib's compress() with the stream kept per thread: compress() is deflateInit + deflate(Z_FINISH) + deflateEnd, and for the 4 KB
 * sections of a bigWig the 268 KB of state it allocates and clears every time are a third of its time (12 800 sections: 1.08 s
 * with compress(), 0.69 s with deflateReset on one core — the bytes are the same, the parameters being compress()'s). */
static size_t z_compress(const void *src, size_t n, void *dst, size_t cap)
{
    static __thread z_stream z;
    static __thread int z_on;
    if (!z_on) {
        memset(&z, 0, sizeof z);
        if (deflateInit(&z, Z_DEFAULT_COMPRESSION) != Z_OK) die("Couldn't zCompress %lld bytes", (long long)n);
        z_on = 1;
    } else if (deflateReset(&z) != Z_OK) {
        die("Couldn't zCompress %lld bytes", (long long)n);
    }
    z.next_in = (Bytef *)src;
    z.avail_in = (uInt)n;
    z.next_out = (Bytef *)dst;
    z.avail_out = (uInt)cap;
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) die("Couldn't zCompress %lld bytes", (long long)n);
    return cap - (size_t)z.avail_out;
}

/* ---- cuskent/cirTree.c: the index over items that carry (chromIx, start, end) and a file offset ------------------- */
typedef struct {
    uint32_t start_chrom, start_base, end_chrom, end_base;
    uint64_t start_off, end_off;
    size_t first_child, n_child;       /* children in the level below (unused at the item level) */
} rnode;

typedef struct {
    uint32_t chrom, start, end;
    uint64_t off;
} ritem;

static void rnode_widen(rnode *p, const rnode *e)
{
    if (e->start_chrom < p->start_chrom) {
        p->start_chrom = e->start_chrom;
        p->start_base = e->start_base;
    } else if (e->start_chrom == p->start_chrom && e->start_base < p->start_base) {
        p->start_base = e->start_base;
    }
    if (e->end_chrom > p->end_chrom) {
        p->end_chrom = e->end_chrom;
        p->end_base = e->end_base;
    } else if (e->end_chrom == p->end_chrom && e->end_base > p->end_base) {
        p->end_base = e->end_base;
    }
}

/* cuskent/cirTree.c:338-367 with rTreeFromChromRangeArray (:141-262) and writeTreeToOpenFile (:264-336) */
static void cir_tree_write(FILE *f, const ritem *it, uint64_t n_items, uint32_t block_size, uint32_t items_per_slot, uint64_t end_file_offset)
{
    /* level 0 here = the slots (what the reference calls the first level above the leaves' items) */
    size_t n0 = (size_t)((n_items + items_per_slot - 1) / items_per_slot);
    rnode *cur = xcalloc(n0 ? n0 : 1, sizeof *cur);
    for (size_t k = 0; k < n0; k++) {
        const uint64_t i = (uint64_t)k * items_per_slot;
        uint64_t one = n_items - i;
        const int final = one <= items_per_slot;
        if (!final) one = items_per_slot;
        rnode *el = &cur[k];
        el->start_chrom = el->end_chrom = it[i].chrom;
        el->start_base = it[i].start;
        el->end_base = it[i].end;
        el->start_off = it[i].off;
        el->end_off = final ? end_file_offset : it[i + one].off;
        for (uint64_t j = 1; j < one; j++) {
            rnode key = {it[i + j].chrom, it[i + j].start, it[i + j].chrom, it[i + j].end, 0, 0, 0, 0};
            rnode_widen(el, &key);
        }
    }
    /* condense until one node is left, at least once (cirTree.c:206-259): levels[0] = root ... levels[count-1] = slots */
    rnode *levels[64];
    size_t level_n[64];
    int count = 1;
    levels[0] = cur;
    level_n[0] = n0;
    while (level_n[0] > 1 || count < 2) {
        const size_t nc = level_n[0];
        const size_t np = (nc + block_size - 1) / block_size;
        rnode *par = xcalloc(np ? np : 1, sizeof *par);
        for (size_t k = 0; k < np; k++) {
            const size_t a = k * block_size, b = a + block_size < nc ? a + block_size : nc;
            par[k] = levels[0][a];
            par[k].first_child = a;
            par[k].n_child = b - a;
            for (size_t j = a + 1; j < b; j++) rnode_widen(&par[k], &levels[0][j]);
        }
        memmove(levels + 1, levels, sizeof(levels[0]) * (size_t)count);
        memmove(level_n + 1, level_n, sizeof(level_n[0]) * (size_t)count);
        levels[0] = par;
        level_n[0] = np;
        count++;
    }
    const rnode *root = &levels[0][0];
    const uint32_t magic = CIR_SIG, reserved = 0;
    PUT(f, magic);
    PUT(f, block_size);
    PUT(f, n_items);
    PUT(f, root->start_chrom);
    PUT(f, root->start_base);
    PUT(f, root->end_chrom);
    PUT(f, root->end_base);
    PUT(f, end_file_offset);
    PUT(f, items_per_slot);
    PUT(f, reserved);
    /* offsets of the levels: every level is priced as index nodes, the leaves included (cirTree.c:276-285) */
    const uint64_t i_node = 4 + 24 * (uint64_t)block_size, l_node = 4 + 32 * (uint64_t)block_size;
    uint64_t level_off[64], off = (uint64_t)ftello(f);
    for (int i = 0; i < count; i++) {
        level_off[i] = off;
        off += level_n[i] * i_node;
    }
    /* index levels 0 .. count-3: nodes whose slots point at the nodes of the next level */
    const int final_level = count - 3;
    for (int i = 0; i <= final_level; i++) {
        const uint64_t child_size = i == final_level ? l_node : i_node;
        uint64_t child_off = level_off[i + 1];
        for (size_t k = 0; k < level_n[i]; k++) {
            const rnode *nd = &levels[i][k];
            const uint8_t is_leaf = 0, res8 = 0;
            const uint16_t cnt = (uint16_t)nd->n_child;
            PUT(f, is_leaf);
            PUT(f, res8);
            PUT(f, cnt);
            for (size_t c = 0; c < nd->n_child; c++) {
                const rnode *el = &levels[i + 1][nd->first_child + c];
                PUT(f, el->start_chrom);
                PUT(f, el->start_base);
                PUT(f, el->end_chrom);
                PUT(f, el->end_base);
                PUT(f, child_off);
                child_off += child_size;
            }
            put_zero(f, 24 * (size_t)(block_size - cnt));
        }
        if ((uint64_t)ftello(f) != level_off[i + 1]) die("Internal error: offset mismatch in the bigWig index");
    }
    /* leaves: the nodes of level count-2, their slots are the elements of the last level; empty slots are padded with
     * 24 bytes each, not 32 (cirTree.c:119-122) */
    const int leaf_level = count - 2;
    for (size_t k = 0; k < level_n[leaf_level]; k++) {
        const rnode *nd = &levels[leaf_level][k];
        const uint8_t is_leaf = 1, res8 = 0;
        const uint16_t cnt = (uint16_t)nd->n_child;
        PUT(f, is_leaf);
        PUT(f, res8);
        PUT(f, cnt);
        for (size_t c = 0; c < nd->n_child; c++) {
            const rnode *el = &levels[leaf_level + 1][nd->first_child + c];
            PUT(f, el->start_chrom);
            PUT(f, el->start_base);
            PUT(f, el->end_chrom);
            PUT(f, el->end_base);
            PUT(f, el->start_off);
            const uint64_t size = el->end_off - el->start_off;
            PUT(f, size);
        }
        put_zero(f, 24 * (size_t)(block_size - cnt));
    }
    for (int i = 0; i < count; i++) free(levels[i]);
}

/* ---- cuskent/bPlusTree.c:419-576 over the chromosome array (key = name padded with zeros, value = id, size) -------- */
static void bpt_write(FILE *f, const bw_chrom *chroms, uint64_t n, uint32_t block_size, uint32_t key_size)
{
    const uint32_t magic = BPT_SIG, reserved = 0, val_size = 8;
    PUT(f, magic);
    PUT(f, block_size);
    PUT(f, key_size);
    PUT(f, val_size);
    PUT(f, n);
    PUT(f, reserved);
    PUT(f, reserved);
    uint64_t index_offset = (uint64_t)ftello(f);
    int levels = 1;
    for (uint64_t c = n; c > block_size; c = (c + block_size - 1) / block_size) levels++;
    char *key = xcalloc(key_size + 1, 1);
    for (int level = levels - 1; level > 0; level--) {
        uint64_t slot_per = 1;
        for (int i = 0; i < level; i++) slot_per *= block_size;
        const uint64_t node_per = slot_per * block_size;
        const uint64_t node_count = (n + node_per - 1) / node_per;
        const uint64_t in_index = 4 + (uint64_t)block_size * (key_size + 8), in_leaf = 4 + (uint64_t)block_size * (key_size + val_size);
        const uint64_t next_block = level == 1 ? in_leaf : in_index;
        /* the reference passes the running offset through a 32-bit parameter (bPlusTree.c:431); a chromosome tree
         * never gets there */
        uint64_t next_child = (uint64_t)(uint32_t)index_offset + node_count * in_index;
        for (uint64_t i = 0; i < n; i += node_per) {
            uint64_t count_one = (n - i + slot_per - 1) / slot_per;
            if (count_one > block_size) count_one = block_size;
            const uint8_t is_leaf = 0, res8 = 0;
            const uint16_t c16 = (uint16_t)count_one;
            PUT(f, is_leaf);
            PUT(f, res8);
            PUT(f, c16);
            uint64_t end_ix = i + node_per;
            if (end_ix > n) end_ix = n;
            for (uint64_t j = i; j < end_ix; j += slot_per) {
                memset(key, 0, key_size);
                strcpy(key, chroms[j].name);
                put(f, key, key_size);
                PUT(f, next_child);
                next_child += next_block;
            }
            put_zero(f, (size_t)(block_size - count_one) * (key_size + 8));
        }
        index_offset = (uint64_t)ftello(f);
    }
    uint64_t left = n;
    for (uint64_t i = 0; i < n;) {
        const uint16_t count_one = (uint16_t)(left > block_size ? block_size : left);
        const uint8_t is_leaf = 1, res8 = 0;
        PUT(f, is_leaf);
        PUT(f, res8);
        PUT(f, count_one);
        for (uint16_t j = 0; j < count_one; j++) {
            memset(key, 0, key_size);
            strcpy(key, chroms[i + j].name);
            put(f, key, key_size);
            PUT(f, chroms[i + j].id);
            PUT(f, chroms[i + j].size);
        }
        put_zero(f, (size_t)(block_size - count_one) * (key_size + val_size));
        left -= count_one;
        i += count_one;
    }
    free(key);
}

/* cuskent/bbiWrite.c:478-536: the zoom records, itemsPerSlot to a deflated block, then their R tree */
static uint64_t write_summary_and_index(FILE *f, bw_sumlist *L, uint32_t block_size, uint32_t items_per_slot)
{
    const uint32_t count = (uint32_t)L->n;
    PUT(f, count);
    const size_t unc_cap = 32 * (size_t)items_per_slot, comp_cap = z_buf_size(unc_cap);
    ritem *items = xcalloc(count ? count : 1, sizeof *items);
    /* slots are deflated in parallel into memory, then laid down in order */
    const size_t n_slots = ((size_t)count + items_per_slot - 1) / items_per_slot, group = 1024;
    char *comp = xmalloc(comp_cap * group);
    size_t *csize = xcalloc(group, sizeof *csize);
    for (size_t g0 = 0; g0 < n_slots; g0 += group) {
        const size_t g1 = g0 + group < n_slots ? g0 + group : n_slots;
#pragma omp parallel for schedule(dynamic, 4)
        for (long sl = (long)g0; sl < (long)g1; sl++) {
            const uint32_t i = (uint32_t)sl * items_per_slot;
            const uint32_t in_slot = count - i > items_per_slot ? items_per_slot : count - i;
            char *unc = xmalloc(unc_cap), *w = unc;
            for (uint32_t k = 0; k < in_slot; k++) {
                const bw_summary *s = &L->v[i + k];
                memcpy(w, &s->chrom_id, 4); w += 4;
                memcpy(w, &s->start, 4); w += 4;
                memcpy(w, &s->end, 4); w += 4;
                memcpy(w, &s->valid_count, 4); w += 4;
                memcpy(w, &s->min_val, 4); w += 4;
                memcpy(w, &s->max_val, 4); w += 4;
                memcpy(w, &s->sum_data, 4); w += 4;
                memcpy(w, &s->sum_squares, 4); w += 4;
            }
            csize[sl - (long)g0] = z_compress(unc, (size_t)(w - unc), comp + (size_t)(sl - (long)g0) * comp_cap, comp_cap);
            free(unc);
        }
        for (size_t sl = g0; sl < g1; sl++) {
            const uint32_t i = (uint32_t)sl * items_per_slot;
            const uint32_t in_slot = count - i > items_per_slot ? items_per_slot : count - i;
            const uint64_t pos = (uint64_t)ftello(f);
            for (uint32_t k = 0; k < in_slot; k++) {
                bw_summary *s = &L->v[i + k];
                s->file_offset = pos;
                items[i + k].chrom = s->chrom_id;
                items[i + k].start = s->start;
                items[i + k].end = s->end;
                items[i + k].off = pos;
            }
            put(f, comp + (sl - g0) * comp_cap, csize[sl - g0]);
        }
    }
    free(csize);
    const uint64_t index_offset = (uint64_t)ftello(f);
    cir_tree_write(f, items, count, block_size, items_per_slot, index_offset);
    free(items);
    free(comp);
    return index_offset;
}

static int cmp_chrom_name(const void *a, const void *b)
{
    return strcmp(((const bw_chrom *)a)->name, ((const bw_chrom *)b)->name);
}

/* names[i] / len[i] / val[i]: the wig blocks (only names with len != 0 belong here, generic.c:83-90); a value is the
 * float the converter makes of the wig's text: (float)strtod("%u" or "%.4f" of the number) */
#include <time.h>
static __thread double bw_t0;
static void bw_tick(const char *what)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    const double t = (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    if (getenv("ITX_TIMING_BW")) fprintf(stderr, "[itx timing] bw: %s %.3f s\n", what, t - bw_t0);
    bw_t0 = t;
}
#define BW_T(x) bw_tick(x)

void write_bigwig(const char *path, const char *wig_name, const char *const *names, const uint32_t *len, const float *const *val,
                  size_t n_names)
{
    const uint32_t block_size = 256, items_per_slot = 1024;                  /* stat.c:157-158 */
    BW_T("start");
    if (n_names == 0) die("%s is empty of data", wig_name);                  /* bwgCreate.c:1108-1109 */
    /* chromosomes in strcmp order (the section sort, bwgCreate.c:138-151), ids in that order (:584-627) */
    bw_chrom *chroms = xcalloc(n_names, sizeof *chroms);
    size_t *src = xcalloc(n_names, sizeof *src);
    for (size_t i = 0; i < n_names; i++) {
        chroms[i].name = names[i];
        chroms[i].id = (uint32_t)i;                                          /* borrowed as the source index until sorted */
        chroms[i].size = len[i];
    }
    qsort(chroms, n_names, sizeof *chroms, cmp_chrom_name);
    uint32_t max_name = 0;
    size_t n_sec = 0;
    for (size_t i = 0; i < n_names; i++) {
        src[i] = chroms[i].id;
        chroms[i].id = (uint32_t)i;
        const uint32_t l = (uint32_t)strlen(chroms[i].name);
        if (l > max_name) max_name = l;
        n_sec += (chroms[i].size + items_per_slot - 1) / items_per_slot;
    }
    bw_section *sec = xcalloc(n_sec ? n_sec : 1, sizeof *sec);
    size_t *sec_of = xcalloc(n_names + 1, sizeof *sec_of);
    size_t k = 0;
    uint64_t full_size = 0;
    for (size_t i = 0; i < n_names; i++) {
        sec_of[i] = k;
        for (uint32_t s = 0; s < chroms[i].size; s += items_per_slot) {
            const uint32_t c = chroms[i].size - s > items_per_slot ? items_per_slot : chroms[i].size - s;
            sec[k].chrom_id = (uint32_t)i;
            sec[k].start = s;
            sec[k].end = s + c;
            sec[k].item_count = c;
            sec[k].val = val[src[i]] + s;
            full_size += 24 + 4 * (uint64_t)c;
            k++;
        }
    }
    sec_of[n_names] = k;

    BW_T("sections");
    /* zoom levels (bwgCreate.c:826-885): step 1 everywhere, so the average resolution is 1 and the first try is 10 */
    int initial_reduction = 1 * 10;
    const uint64_t max_reduced = full_size / 2;
    uint64_t last_summary_size = 0;
    bw_sumlist sums[10];
    uint32_t reductions[10];
    for (;;) {
        /* How many summaries a reduction gives needs no arithmetic on the values: every sequence is covered base by base from
         * 0 to its size, so its summaries tile it in steps of the reduction (bbiWrite.c:381-395). The values are reduced once,
         * with the reduction the loop settles on. */
        uint64_t n_first = 0;
        for (size_t c = 0; c < n_names; c++) n_first += ((uint64_t)chroms[c].size + (uint64_t)initial_reduction - 1) / (uint64_t)initial_reduction;
        uint64_t size = n_first * 32;
        size *= 2;                                                             /* "summary not compressing as well as primary data" */
        if (size >= max_reduced && size != last_summary_size) {
            int next = (int)(1.1 * initial_reduction * (double)size / (double)max_reduced);
            if (next < initial_reduction * 2) next = initial_reduction * 2;
            initial_reduction = next;
            last_summary_size = size;
        } else {
            sums[0] = reduce_sections(sec, sec_of, chroms, n_names, initial_reduction);
            if ((uint64_t)sums[0].n != n_first) die("internal error: bigWig first zoom level has %zu summaries, %llu expected", sums[0].n, (unsigned long long)n_first);
            break;
        }
    }
    int n_sums = 1;
    reductions[0] = (uint32_t)initial_reduction;
    uint64_t reduction = (uint64_t)initial_reduction;
    for (int i = 0; i < 9; i++) {
        reduction *= 4;
        if (reduction > 1000000000) break;
        bw_sumlist L = reduce_summaries(&sums[n_sums - 1], chroms, n_names, (int)reduction);
        const uint64_t size = (uint64_t)L.n * 32;
        const size_t items = L.n;
        if (size != last_summary_size) {
            sums[n_sums] = L;
            reductions[n_sums] = (uint32_t)reduction;
            n_sums++;
        } else
            free(L.v);
        if (items <= n_names) break;
    }

    BW_T("zoom lists");
    FILE *f = fopen(path, "wb");
    if (!f) die("mustOpen: Can't open %s to write: %s", path, strerror(errno));
    const uint32_t sig = BW_SIG, res32 = 0;
    const uint16_t version = BW_VERSION, summary_count = (uint16_t)n_sums, res16 = 0;
    const uint64_t res64 = 0;
    uint64_t chrom_tree_off = 0, data_off = 0, index_off = 0, total_summary_off = 0;
    uint32_t unc_buf = 0;
    PUT(f, sig);
    PUT(f, version);
    PUT(f, summary_count);
    const long chrom_tree_pos = ftell(f);
    PUT(f, chrom_tree_off);
    const long data_pos = ftell(f);
    PUT(f, data_off);
    const long index_pos = ftell(f);
    PUT(f, index_off);
    PUT(f, res16);
    PUT(f, res16);
    PUT(f, res64);
    const long total_pos = ftell(f);
    PUT(f, total_summary_off);
    const long unc_pos = ftell(f);
    PUT(f, unc_buf);
    PUT(f, res64);
    long zoom_pos[10];
    for (int i = 0; i < n_sums; i++) {
        PUT(f, reductions[i]);
        PUT(f, res32);
        zoom_pos[i] = ftell(f);
        PUT(f, res64);
        PUT(f, res64);
    }
    total_summary_off = (uint64_t)ftello(f);
    put_zero(f, 40);
    chrom_tree_off = (uint64_t)ftello(f);
    bpt_write(f, chroms, n_names, block_size < n_names ? block_size : (uint32_t)n_names, max_name);
    data_off = (uint64_t)ftello(f);
    const uint64_t section_count = n_sec;
    PUT(f, section_count);
    {
        /* sections are deflated in parallel into memory, then laid down in order (their offsets feed the index) */
        const size_t cap = 24 + 4 * (size_t)items_per_slot, ccap = z_buf_size(cap);
        const size_t group = 4096;
        char *comp = xmalloc(ccap * group);
        size_t *csize = xcalloc(group, sizeof *csize);
        for (size_t g0 = 0; g0 < n_sec; g0 += group) {
            const size_t g1 = g0 + group < n_sec ? g0 + group : n_sec;
            uint32_t unc_max = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(max : unc_max)
            for (long i = (long)g0; i < (long)g1; i++) {
                const bw_section *s = &sec[i];
                const uint32_t step = 1, span = 1;
                const uint8_t type = 3, r8 = 0;                                /* bwgTypeFixedStep */
                const uint16_t cnt = (uint16_t)s->item_count;
                char buf[24 + 4 * 1024];
                char *w = buf;
                memcpy(w, &s->chrom_id, 4); w += 4;
                memcpy(w, &s->start, 4); w += 4;
                memcpy(w, &s->end, 4); w += 4;
                memcpy(w, &step, 4); w += 4;
                memcpy(w, &span, 4); w += 4;
                memcpy(w, &type, 1); w += 1;
                memcpy(w, &r8, 1); w += 1;
                memcpy(w, &cnt, 2); w += 2;
                for (uint32_t j = 0; j < s->item_count; j++) {
                    memcpy(w, &s->val[j], 4);
                    w += 4;
                }
                const uint32_t unc = (uint32_t)(w - buf);
                if (unc > unc_max) unc_max = unc;
                csize[i - (long)g0] = z_compress(buf, unc, comp + (size_t)(i - (long)g0) * ccap, ccap);
            }
            if (unc_max > unc_buf) unc_buf = unc_max;
            for (size_t i = g0; i < g1; i++) {
                sec[i].file_offset = (uint64_t)ftello(f);
                put(f, comp + (i - g0) * ccap, csize[i - g0]);
            }
        }
        free(comp);
        free(csize);
    }
    BW_T("data");
    index_off = (uint64_t)ftello(f);
    {
        ritem *items = xcalloc(n_sec ? n_sec : 1, sizeof *items);
        for (size_t i = 0; i < n_sec; i++) {
            items[i].chrom = sec[i].chrom_id;
            items[i].start = sec[i].start;
            items[i].end = sec[i].end;
            items[i].off = sec[i].file_offset;
        }
        cir_tree_write(f, items, n_sec, block_size, 1, index_off);
        free(items);
    }
    BW_T("index");
    uint64_t zoom_data[10], zoom_index[10];
    for (int i = 0; i < n_sums; i++) {
        zoom_data[i] = (uint64_t)ftello(f);
        zoom_index[i] = write_summary_and_index(f, &sums[i], block_size, items_per_slot);
    }
    BW_T("zooms");
    /* the file-wide summary from the first zoom level (bwgCreate.c:966-988) */
    if (sums[0].n) {
        const bw_summary *s = &sums[0].v[0];
        uint64_t valid = s->valid_count;
        double mn = s->min_val, mx = s->max_val, sd = s->sum_data, sq = s->sum_squares;
        for (size_t i = 1; i < sums[0].n; i++) {
            s = &sums[0].v[i];
            valid += s->valid_count;
            if (s->min_val < mn) mn = s->min_val;
            if (s->max_val > mx) mx = s->max_val;
            sd += s->sum_data;
            sq += s->sum_squares;
        }
        fseeko(f, (off_t)total_summary_off, SEEK_SET);
        PUT(f, valid);
        PUT(f, mn);
        PUT(f, mx);
        PUT(f, sd);
        PUT(f, sq);
    } else
        total_summary_off = 0;
    fseek(f, data_pos, SEEK_SET);
    PUT(f, data_off);
    fseek(f, index_pos, SEEK_SET);
    PUT(f, index_off);
    fseek(f, chrom_tree_pos, SEEK_SET);
    PUT(f, chrom_tree_off);
    fseek(f, total_pos, SEEK_SET);
    PUT(f, total_summary_off);
    if (32 * items_per_slot > unc_buf) unc_buf = 32 * items_per_slot;
    fseek(f, unc_pos, SEEK_SET);
    PUT(f, unc_buf);
    for (int i = 0; i < n_sums; i++) {
        fseek(f, zoom_pos[i], SEEK_SET);
        PUT(f, zoom_data[i]);
        PUT(f, zoom_index[i]);
    }
    fseek(f, 0L, SEEK_END);
    PUT(f, sig);
    if (fclose(f) != 0) die("carefulClose: error closing %s", path);
    for (int i = 0; i < n_sums; i++) free(sums[i].v);
    free(sec);
    free(sec_of);
    free(src);
    free(chroms);
}
