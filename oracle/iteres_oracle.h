/* iteres_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the one hot path of lidaof/iteres: per BAM
 * record -> record filter + coordinate derivation -> binKeeperFind -> best-hit
 * rule -> counters / per-base coverage (stat) or per-locus counts (filter).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (iteres_amd/) never does.
 *
 * Parity status: PINNED. tests/test_oracle_golden.py checks this restatement
 * against outputs of the reference itself (oracle/_ref/iteres, compiled from
 * /root/reference by oracle/Makefile) committed under tests/golden/.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#ifndef ITERES_ORACLE_H
#define ITERES_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_table orc_table;

typedef struct {
    uint32_t mapq_min;            /* -Q   (stat.c:49, filter.c:49)   */
    float    min_cov;             /* -c / -g (stat.c:50, filter.c:50) */
    uint32_t extension;           /* -E   (stat.c:61)                */
    uint32_t isize_max;           /* -I   (stat.c:62)                */
    int32_t  treat_pe_as_se;      /* -T   (stat.c:55)                */
    int32_t  discard_half_mapped; /* -D   (stat.c:56)                */
    int32_t  filter_mode;         /* 0: stat accumulate (generic.c:983-1024); 1: per-locus (generic.c:662-666) */
} orc_params;

/* rep_len[r] = consensus length from the repeat-size file, 0 when absent (generic.c:1647). */
orc_table *orc_table_new(int n_chrom, const int64_t *chrom_size, int n_rep, const uint32_t *rep_len,
                         int n_fam, int n_cla);
void orc_table_free(orc_table *t);

/* One rmsk row in file order (generic.c:1592-1626 + cuskent/binRange.c:171-186).
 * Returns the row index (>= 0) it was stored under, -1 if the row is dropped because its
 * chromosome is not in the chrom-size file (generic.c:1618-1622), -2 for the conditions on
 * which binKeeperAdd errAborts. */
int64_t orc_table_add(orc_table *t, int chrom, uint32_t start, uint32_t end, uint32_t cons_start,
                      uint32_t cons_end, uint32_t rep, uint32_t fam, uint32_t cla);

/* orc_table_add over n rows in file order; out[i] = orc_table_add's return for row i. */
void orc_table_add_many(orc_table *t, size_t n, const int32_t *chrom, const uint32_t *start, const uint32_t *end,
                        const uint32_t *cons_start, const uint32_t *cons_end, const uint32_t *rep,
                        const uint32_t *fam, const uint32_t *cla, int64_t *out);

/* cuskent/binRange.c:196-227. Writes at most cap row indices in the order of the list the
 * reference returns; returns the total number of hits. */
int64_t orc_find(const orc_table *t, int chrom, int start, int end, int64_t *rows, int64_t cap);

/* Offsets of each repName's coverage vector inside the concatenated cov arrays:
 * off[r] = sum_{q<r} rep_len[q]; off[n_rep] = total. */
void orc_cov_offsets(const orc_table *t, uint64_t *off);

/* The per-record loop, generic.c:745-1036 (stat copy) == generic.c:385-697 (filter copy).
 * Inputs are BAM core fields per record; tmpend[i] is `n_cigar ? bam_calend : pos + l_qseq`
 * (generic.c:820). tid2chrom[tid]: chrom index, -1 when the (possibly -C renamed) name is not in
 * the chrom-size file (generic.c:793-801), -2 when -C drops it ("GL*", generic.c:783-784).
 * Outputs (all accumulated into, caller zeroes): hit_row[i] = chosen rmsk row or -1;
 * cnt[13] (generic.c:1048-1060); rep/fam/cla counters [0..n) all reads, [n..2n) unique reads;
 * cov / cov_uniq per orc_cov_offsets; locus_cnt[row] (filter mode, slCount of ss->sl).
 * skip (may be NULL): skip[r] != 0 marks a record the reference would leave with one of the two `continue`s
 * between the mapped-read counters and the accumulation — a -R duplicate (generic.c:907-919) or an XA veto
 * (generic.c:972-982), both decided by string logic outside this restatement; such a record is counted up to
 * cnt[11] and goes no further. */
int orc_run(const orc_table *t, const orc_params *p, int n_tid, const int32_t *tid2chrom, size_t n,
            const int32_t *tid, const int32_t *pos, const int32_t *tmpend, const uint8_t *mapq,
            const uint16_t *flag, const int32_t *mpos, const int32_t *isize, const uint8_t *skip, int64_t *hit_row,
            uint64_t *cnt, uint64_t *rep_cnt, uint64_t *fam_cnt, uint64_t *cla_cnt, uint32_t *cov,
            uint32_t *cov_uniq, uint32_t *locus_cnt);

/* cuskent/hash.c:41-53 */
uint32_t orc_hash_string(const char *s);

#ifdef __cplusplus
}
#endif
#endif
