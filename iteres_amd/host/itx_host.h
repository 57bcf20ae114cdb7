/* itx_host.h — the C host side of the drop-in: `iteres stat` / `iteres filter` with the reference's command
 * line, stderr banners, exit codes and output file formats (stat.c, filter.c, generic.c:53-113,1709-1746 of
 * /root/reference), driving the MI355X engine through include/iteres_amd.h. Everything here is host logic:
 * parsing, name bookkeeping in the reference's hash iteration order, BAM/SAM decoding into the pinned record
 * SoA, and the writers. There is no CPU implementation of the hot path in this program. */
#ifndef ITX_HOST_H
#define ITX_HOST_H
#include <stdint.h>
#include <stdio.h>
#include <stddef.h>

#include "../../include/iteres_amd.h"

#define ITERES_VERSION "0.3.3-r123"      /* generic.h:4 of the reference: same version string in the usage text */

/* ---- errors: cuskent/errabort.c:166-199 — message + newline to stderr, exit(-1) */
void die(const char *fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));
void warnf(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
extern FILE *itx_err_stream;             /* where die / warnf write when not stderr (ranks > 0 of a multi-GPU job) */
extern const char *itx_err_prefix;
extern void (*itx_die_hook)(void);        /* run by die() before it leaves (rank 0 ends the ranks it started) */
void *xmalloc(size_t n);
void *xcalloc(size_t n, size_t sz);
void *xrealloc(void *p, size_t n);
char *xstrdup(const char *s);

/* ---- string -> dense id table that remembers insertion order and can list its names in the iteration
 * order of a kent hash (cuskent/hash.c:41-53,115-142,374-410,511-552) built by the same insertions. */
typedef struct {
    char **name;
    uint32_t n, cap;
    uint32_t *bucket;      /* open hashing for lookups */
    uint32_t *next;
    uint32_t nbucket;
} names_t;
void names_init(names_t *t);
void names_free(names_t *t);
int64_t names_find(const names_t *t, const char *s);
uint32_t names_intern(names_t *t, const char *s);             /* id of s, adding it when new */
uint32_t kent_hash_string(const char *s);                     /* cuskent/hash.c:41-53 */
/* order[i] = id of the i-th name hashFirst/hashNext would visit (hash started with 2^start_pow2 buckets). */
void names_kent_order(const names_t *t, int start_pow2, uint32_t *order);

/* ---- two-column "name value" files (cuskent/obscure.c:139-150 hashNameIntFile): later duplicates win */
typedef struct {
    names_t names;
    int64_t *value;
} sizes_t;
void sizes_load(const char *path, sizes_t *out);
void sizes_free(sizes_t *s);
/* value of name or dflt (hashIntValDefault, cuskent/hash.c:250-258) */
int64_t sizes_get(const sizes_t *s, const char *name, int64_t dflt);

/* ---- rmsk.txt -> rows + name tables (generic.c:1578-1707 without the binKeeper) */
typedef struct {
    names_t chroms;            /* chromosomes that got a binKeeper, in first-seen order: key order of hashRmsk   */
    names_t reps, fams, clas;  /* hashRep / hashFam / hashCla insertion order                                     */
    itx_row *rows;             /* kept rows, file order; .chrom indexes chrom_size                                */
    uint32_t *row_chrom_name;  /* [n_rows] index into chroms                                                      */
    size_t n_rows;
    int64_t *chrom_size;       /* [chroms.n] size from the chrom-size file                                        */
    uint32_t *rep_len;         /* [reps.n]  repeat-size file value or 0 (generic.c:1647)                          */
    uint32_t *rep_fam, *rep_cla;   /* family / class string ids of the FIRST row of each name (generic.c:1638-1640) */
    uint32_t *fam_cla;         /* class of the first row of each family (generic.c:1667-1668)                     */
    uint64_t *rep_genome, *rep_total, *fam_genome, *fam_total, *cla_genome, *cla_total;   /* genome_count, total_length */
    int repeat_num;            /* rows counted by the banner (generic.c:1593,1697)                                */
} rmsk_t;
void rmsk_load(const char *path, const sizes_t *chr_sizes, const sizes_t *rep_sizes, int filter_field, const char *filter_name,
               rmsk_t *out);
void rmsk_free(rmsk_t *r);

/* ---- alignment input: BAM (BGZF) or SAM text, decoded into the engine's record SoA */
typedef struct aln_reader aln_reader;
/* BAM decoding on the device (include/iteres_amd.h: itx_bamwin_*): BGZF blocks inflated, records located and parsed there;
 * the reader only moves compressed bytes in and the per-record arrays out. The reader stays free of any link-time
 * dependency on the HIP library: the driver hands it the entry points before it opens a file. `alloc` / `release`:
 * page-locked memory for the compressed chunks. NULL restores the host decoder. */
typedef struct aln_device_ops {
    itx_inflater *ctx;
    int (*push_begin)(itx_inflater *, int, int, const void *, size_t, const itx_bgzf_block *, size_t);
    int (*push_end)(itx_inflater *, int, uint8_t *, size_t *);
    int (*patch)(itx_inflater *, int, size_t, const void *, size_t);
    int (*truncate)(itx_inflater *, int, size_t);
    int (*carry)(itx_inflater *, int, int);
    int (*avail)(const itx_inflater *, int, size_t *);
    int (*peek)(itx_inflater *, int, size_t, void *, size_t);
    int (*skip)(itx_inflater *, int, size_t);
    int (*parse)(itx_inflater *, int, int, size_t *, int *, int *, size_t *);
    int (*fetch)(itx_inflater *, size_t, size_t, const itx_staging *, size_t, uint32_t *, uint8_t *);
    int (*bytes)(itx_inflater *, size_t, void *, size_t);
    int (*tids)(itx_inflater *, uint8_t *, int);
    int (*device_batch)(itx_inflater *, size_t, int, itx_batch *);
    void *(*alloc)(size_t bytes);
    void (*release)(void *p);
    const char *(*last_error)(void);
    /* what the driver reserved on the device (itx_inflater_reserve): windows to rotate through, blocks / inflated bytes per push;
     * zeros: the defaults (ITX_BAMWIN_WINDOWS, 16384 blocks, 1 GiB) with buffers that grow on demand */
    int n_windows;
    size_t max_blocks, max_bytes;
    int (*xa_veto)(itx_inflater *, itx_xaveto *, size_t, size_t, uint64_t *, uint64_t *);      /* itx_bamwin_xa_veto */
    int (*push_copied)(itx_inflater *, int);       /* itx_bamwin_push_copied: lane s's compressed bytes have left the caller's buffer */
} aln_device_ops;
void aln_use_device(const aln_device_ops *ops);
size_t aln_raw_step(size_t left);           /* bytes per read step of a regular file with `left` bytes to go (after aln_use_device) */
/* compressed bytes per chunk handed to the device decoder (ITX_BGZF_CHUNK overrides). Pass 1 of the decoder is a lane per
 * block and bound by the latency of one block's symbol chain (~29 ms for a block of literal-heavy content, whatever the number
 * of blocks, until every SIMD holds a wave: 65 k blocks): its throughput is the blocks in flight. Blocks of real BAM content
 * take ~33 KB compressed, so 384 MB = 12 k blocks per push, four pushes in flight = 48 k blocks (round 2: 128 MB, written for
 * blocks of 15 KB). Measured, 200 M reads of 40-value-quality content: record loop 1.35 s -> 0.64 s. */
#define ALN_DEVICE_CHUNK (384u << 20)
#define ALN_DEVICE_RAW_BUFFERS 5             /* page-locked chunk buffers the reader of a regular file rotates through (bamio.c) */
/* Device decoder only. aln_device_window: 1 when the next records can be taken as DEVICE arrays — the reader stands at
 * the start of a decoded window (decoding the next one if need be); *flags: bit 0 some record of the window is paired,
 * bit 1 some record carries an XA tag; *tid_seen[n_targets]: references with a mapped record in the window. 0: host
 * decoder, end of input, or in the middle of a window: use aln_read_batch (which then stops at the window's end).
 * aln_read_batch_device: the next <= cap records of that window as device arrays, valid until the next reader call. */
int aln_device_window(aln_reader *r, int *flags, const uint8_t **tid_seen);
/* Device decoder only: fn(ctx, n_rec) is called once for every window right after its records have been located and parsed, in
 * file order, before any of them is handed out (-R on the device marks its duplicates there: stream.c) */
void aln_set_window_hook(aln_reader *r, void (*fn)(void *ctx, size_t n_rec), void *ctx);
size_t aln_read_batch_device(aln_reader *r, size_t cap, itx_batch *b);
/* the XA veto over the batch aln_read_batch_device has just handed out (its chosen rows are in the veto object's buffer);
 * aln_device_rewind: hand the current window's records out again from its first one (the host route after all) */
int aln_device_xa_veto(aln_reader *r, itx_xaveto *x, size_t n, uint64_t *n_vetoed, uint64_t *n_hard);
void aln_device_rewind(aln_reader *r, size_t n);       /* the last n records handed out by aln_read_batch_device are handed out again (by whichever route reads next) */
size_t aln_device_left(const aln_reader *r);  /* device decoder: records of the current window not yet taken */
int aln_device_exhausted(aln_reader *r);
void aln_readahead(aln_reader *r);            /* BAM: start decoding ahead of the first aln_read_batch */     /* device decoder: 1 when no record is left */
aln_reader *aln_open(const char *path, int is_sam);          /* NULL when the file cannot be opened / has no header */
/* One rank's share of a BAM file (multi-GPU; device decoder): the records that start between the split points of the
 * compressed byte offsets lo and hi (hi = SIZE_MAX: to the end; (0, SIZE_MAX) is aln_open). aln_range_verified, after the
 * last batch: 1 when the share's end boundary proved to be a true record start (or the share runs to the end of the file). */
aln_reader *aln_open_range(const char *path, size_t lo, size_t hi);
int aln_range_verified(const aln_reader *r);
/* the split point at or after compressed byte `at` (1 found: BGZF block at file offset *block, *off inflated bytes into it,
 * the block takes *csize bytes; 0 none before the end of the file; -1 the file cannot be read) */
int aln_find_split(const char *path, size_t at, size_t *block, size_t *off, size_t *csize);
void aln_close(aln_reader *r);
int aln_n_targets(const aln_reader *r);
const char *aln_target_name(const aln_reader *r, int tid);
/* What the decoder keeps per record beside the SoA, when the caller wants it (arrays of the batch capacity). */
typedef struct {
    int want_qnames, want_aux;
    int has_strings;           /* set by the reader when the batch left any string behind (else every entry is still NULL)       */
    char **qname;              /* malloc'd copies of the read names (filter -r, stat -B/-V)                              */
    char **xa;                 /* malloc'd value of the XA:Z tag, NULL when the record has none (bam_aux2Z, bam_aux.c:193) */
    int32_t *nm;               /* NM:i, 0 when absent (bam_aux2i, bam_aux.c:159-170)                                      */
} aln_side;
/* Fills up to cap records of the staging slot; returns the number read (0 at end of input).
 * side may be NULL. *any_paired is set when a record with the PAIRED flag was seen; *aux_xa when one carries an XA tag. */
size_t aln_read_batch(aln_reader *r, itx_staging *st, size_t cap, aln_side *side, int *any_paired, int *aux_xa);

/* ---- shared by the two drivers */
typedef struct {
    int is_sam, add_chr, treat, discard, keep_wig, xa_veto, dedup;
    unsigned mapq, isize, extension;
    float min_cov;
    const char *chr_size_file, *rep_size_file, *rmsk_file, *aln_arg;
    const char *bed_path, *bed_uniq_path;      /* stat -B / -V (generic.c:925-936), NULL = off */
} run_opts;
/* counters of the record loop that only the host sees (generic.c:1048-1060) */
typedef struct {
    unsigned long long diff_subfam;            /* cnt[12]: records the XA veto dropped                                  */
    unsigned long long dup_unique;             /* MAPQ >= -Q records -R dropped: cnt[11] = cnt[7] - dup_unique         */
} host_counts;

/* ---- side.c: the string / file-order parts of the record loop */
typedef struct {
    uint32_t start, end;
    char strand;
} host_iv;
/* generic.c:764-905: 1 when the record reaches reads_mapped++ (then *d is its interval), chrom as in the tid map */
int host_derive(const run_opts *o, int32_t chrom, int64_t chrom_size, unsigned flag5, int32_t pos, int32_t tmpend, int32_t mpos, int32_t isize,
                host_iv *d);
typedef struct dup_set dup_set;
dup_set *dup_set_new(void);
void dup_set_free(dup_set *s);
int dup_set_seen(dup_set *s, uint32_t chr_name_id, const host_iv *d, int uniq);      /* generic.c:907-919: 1 = drop */
typedef struct xa_index xa_index;
xa_index *xa_index_new(const rmsk_t *rm);
uint32_t *xa_rep_words(const rmsk_t *rm);            /* [reps.n] equal for names that differ only in case (sameWord); caller frees */
void xa_index_free(xa_index *x);
/* generic.c:303-341: 1 = veto. chosen_rep: repName id of the chosen row; xa is chopped in place. */
int xa_veto(const xa_index *x, uint32_t chosen_rep, int nm, char *xa, int qlen);
/* generic.c:7-15 */
char *filename_without_ext(const char *path);
/* Runs the record loop (generic.c:700-1062 / 343-697) over one or more files through the engine.
 * progress_every: 100000 (stat, generic.c:760) or 10000 (filter, generic.c:397). want_qnames: per-locus read
 * names (filter -r): *locus_names[row] receives a comma-joined list in BAM order. */
/* records parsed ahead of the table by the helper thread (stream.c): allowed when nothing per record is the host's business;
 * the size file's table has to be announced once it is loaded */
void stream_prefetch_allow(const run_opts *o, int filter_mode, int allowed);
void stream_sizes_ready(const sizes_t *chr_sizes);
void gpu_warmup_start(int bam_input, const char *aln_arg, int multi_file, int splittable);  /* starts the HIP runtime on a helper thread (and, for BAM input, the device inflater with its
                                        * page-locked buffers); run_stream joins it */
void run_stream(const run_opts *o, const rmsk_t *rm, const sizes_t *chr_sizes, int filter_mode, int multi_file,
                unsigned progress_every, int want_qnames, itx_engine **eng_out, itx_table **tab_out, char ***locus_names,
                host_counts *hc);

/* writers (generic.c:35-41,53-113,1709-1746) */
double cal_rpkm(unsigned long long reads, unsigned long long total_length, unsigned long long mapped);
double cal_rpm(unsigned long long reads, unsigned long long mapped);
void write_report(const char *path, const uint64_t *cnt, unsigned mapq, const char *subfam);
void write_wig_and_stat(const rmsk_t *rm, const itx_result *res, const uint64_t *cov_off, const char *f_stat, const char *f_wig,
                        const char *f_fam, const char *f_cla, const char *f_wig_uniq, unsigned long long reads_num,
                        unsigned long long reads_num_unique);
void write_filter_out(const rmsk_t *rm, const uint32_t *locus_cnt, char **locus_names, const char *path, int readlist, int threshold,
                      const char *subfam, unsigned long long reads_num);

/* bigwig.c: the bigWig of one set of wig blocks (stat.c:156-158); only names with a consensus length belong here */
void write_bigwig(const char *path, const char *wig_name, const char *const *names, const uint32_t *len, const float *const *val,
                  size_t n_names);
/* tables.c: a whole text file in memory, through the same openers as the rmsk file (plain, .gz/.Z, .bz2, .zip) */
char *slurp_text(const char *path, size_t *len);

/* ---- multi.c: one process per GPU. `iteres stat|filter` starts N - 1 more copies of itself (ranks 1..N-1, GPU r each;
 * N = ITX_GPUS, default: every GPU the process may use, as far as the input is worth sharing; ITX_GPU_MAP=a,b,.. names
 * other devices) before anything touches a GPU; a launcher that starts the ranks
 * itself (bench.py under torch.distributed.run) sets ITX_RANK / ITX_WORLD / ITX_DEVICE / ITX_COMM_ID / ITX_EXCHANGE instead.
 * Every rank parses the same inputs, holds a replica of the table and takes its share of the alignment files' compressed
 * bytes (stream.c); ONE sum-reduce of the engines' partials onto rank 0 ends the stream (include/iteres_amd.h: itx_comm_*),
 * rank 0 writes the files, the others leave. */
void multi_early(int argc, char **argv);            /* from main(): picks up a launcher's rank, quiets ranks > 0 */
void multi_begin(int splittable, const char *aln_arg, int multi_file);   /* after the options are known, before the first GPU call: starts the other ranks */
/* shares.c: this rank's share of every alignment file (compressed byte ranges; aln_open_range turns them into record boundaries) */
typedef struct {
    size_t lo, hi;             /* hi = SIZE_MAX: to the end of the file; lo == hi: nothing of this file */
} share_t;
/* sh[fi] for `rank` of `world`; returns 0 when the job cannot be shared (then rank 0 has everything, the others nothing):
 * not splittable, a file that is no regular file, or less than min_share compressed bytes per rank */
int plan_shares(char **files, int n_files, int splittable, int rank, int world, size_t min_share, share_t *sh);
size_t multi_min_share(void);                       /* compressed bytes below which a share is not worth a rank (ITX_SPLIT_MIN) */
void multi_finish(void);                            /* rank 0, before it returns: the ranks it started have all left */
int multi_rank(void);
int multi_world(void);
int multi_device(void);
int multi_comm_mode(void);
int multi_selftest(void);
const char *multi_comm_id(void);
/* stream.c: what the writers read — itx_engine_finish, or, after a multi-GPU stream, the same from the reduced partial */
int stream_finish(itx_engine *eng, const itx_result *res);

int main_cpgstat(int argc, char **argv);
int main_cpgfilter(int argc, char **argv);

void write_cpg_loci(const rmsk_t *rm, const int *cpg_count, const double *cpg_total, const char *path, const char *subfam, double threshold);

int main_stat(int argc, char **argv);
int main_filter(int argc, char **argv);
/* numa.c */
void numa_place(void);              /* at program start, before any thread: ITX_CPUS, else the processors of the GPU's memory node (ITX_NUMA=0: no) */
void numa_spawn_begin(void);        /* around the start of another rank: it inherits the affinity the process was started with */
void numa_spawn_end(void);

#endif
