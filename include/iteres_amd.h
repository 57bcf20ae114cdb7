/* iteres_amd.h — C ABI of the MI355X (gfx950) engine for iteres' one hot path:
 *
 *     BAM record -> RepeatMasker-interval overlap classification -> per-repName / repFamily /
 *     repClass read counters, per-base consensus coverage (iteres stat) and per-locus read
 *     counts (iteres filter).
 *
 * The reference (lidaof/iteres, /root/reference) has no library or FFI surface: the path is
 * two static loops inside the program. This header is therefore the seam a maintainer would cut
 * INSIDE the reference: every entry point names the reference code it stands in for (paths
 * relative to /root/reference). INTEGRATION.md shows the calls a patched generic.c / stat.c /
 * filter.c would make.
 *
 * Conventions: plain C, no global state, every function returns ITX_OK (0) or a negative ITX_E_*
 * code; itx_last_error() gives a thread-local message. One submitting thread per engine.
 * All "host" pointers are ordinary host memory; "device" pointers are HIP device memory on the
 * engine's GPU. There is no CPU fallback: every call that needs the GPU fails with
 * ITX_E_NO_DEVICE when none is usable.
 */
#ifndef ITERES_AMD_H
#define ITERES_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ITX_OK 0
#define ITX_E_ARG (-1)        /* bad argument */
#define ITX_E_RANGE (-2)      /* a table row is outside its chromosome: what binKeeperAdd errAborts on (cuskent/binRange.c:176-178) */
#define ITX_E_NO_DEVICE (-3)  /* no usable gfx950 device / HIP runtime error */
#define ITX_E_NOMEM (-4)
#define ITX_E_STATE (-5)      /* call order violated (e.g. submit before set_tidmap) */
#define ITX_E_LIMIT (-6)      /* a size exceeds what the device layout encodes */

const char *itx_last_error(void);
/* ABI version of this header: major*1000 + minor. */
int itx_abi_version(void);
/* Number of visible HIP devices, or a negative ITX_E_* code. Does not create a context. */
int itx_device_count(void);

/* ---------------------------------------------------------------------------------------------
 * The repeat table: replaces `struct hash *hashRmsk` (per-chromosome binKeeper of struct rmsk,
 * generic.h:7-15, built by rmsk2binKeeperHash generic.c:1578-1626) together with the per-row
 * links to hashRep / hashFam / hashCla entries (generic.c:1631-1693) and the repeat-size lookup
 * (generic.c:1647).
 * ------------------------------------------------------------------------------------------- */
typedef struct itx_table itx_table;

/* One rmsk.txt row as generic.c:1594-1607 parses it. `chrom` indexes chrom_size[]; rep/fam/cla
 * are dense ids of the row's OWN repName / repFamily / repClass strings (repName -> family is
 * not functional in rmsk, so the ids come from the row). Rows are given in FILE order: the
 * order decides ties exactly as binKeeper's insertion order does (cuskent/binRange.c:185,209-225). */
typedef struct itx_row {
    int32_t  chrom;
    uint32_t start, end;            /* genoStart, genoEnd                        (generic.c:1602-1603) */
    uint32_t cons_start, cons_end;  /* consensus_start / consensus_end           (generic.c:1596-1601) */
    uint32_t rep, fam, cla;
} itx_row;

/* chrom_size[c] = value from the chrom-size file; rep_len[r] = consensus length from the
 * repeat-size file or 0 (generic.c:1647). Fails with ITX_E_RANGE on a row binKeeperAdd would
 * abort on; *bad_row (optional) receives its index. device = HIP device ordinal. */
int itx_table_create(const itx_row *rows, size_t n_rows, const int64_t *chrom_size, int n_chrom,
                     const uint32_t *rep_len, uint32_t n_rep, uint32_t n_fam, uint32_t n_cla,
                     int device, itx_table **out, size_t *bad_row);
void itx_table_destroy(itx_table *t);

typedef struct itx_table_info {
    uint64_t n_rows, n_rep, n_fam, n_cla;
    uint64_t cov_len;      /* sum of rep_len: length of the concatenated coverage vectors          */
    uint64_t n_units;      /* distinct (repName, repFamily, repClass) triples among the rows       */
    uint64_t n_slots;      /* consensus slots the device accumulates over: sum over units of (rep_len + 1) */
    uint64_t table_bytes;  /* device bytes the table occupies                                      */
    int32_t  n_chrom, bin_shift, device, reserved;
} itx_table_info;
int itx_table_get_info(const itx_table *t, itx_table_info *out);
/* off[r] = start of repName r inside cov/cov_uniq; off[n_rep] = cov_len. */
int itx_table_cov_offsets(const itx_table *t, uint64_t *off);

/* ---------------------------------------------------------------------------------------------
 * The engine: replaces the body of the record loop, generic.c:748-1036 (stat copy) ==
 * generic.c:385-697 (filter copy): read-end / mapped / used counters, coordinate derivation,
 * binKeeperFind + best-hit rule (generic.c:945-970), and the accumulate step
 * (generic.c:983-1032). Host-side, order-dependent features stay with the caller: -R dedup
 * (generic.c:907-919), bed emission (925-936), the XA/NM veto (972-982), qname lists (662-666).
 * ------------------------------------------------------------------------------------------- */
typedef struct itx_engine itx_engine;

typedef struct itx_params {
    uint32_t mapq_min;             /* -Q  unique-read MAPQ threshold           (stat.c:49)  */
    float    min_cov;              /* -c / -g coverage threshold               (stat.c:50)  */
    uint32_t extension;            /* -E  0 = none                             (stat.c:61)  */
    uint32_t isize_max;            /* -I                                       (stat.c:62)  */
    int32_t  treat_pe_as_se;       /* -T                                       (stat.c:55)  */
    int32_t  discard_half_mapped;  /* -D                                       (stat.c:56)  */
    int32_t  mode;                 /* ITX_MODE_STAT: counters + coverage; ITX_MODE_FILTER: per-locus counts */
    int32_t  accum;                /* ITX_ACCUM_*: how the device accumulates (same results)            */
} itx_params;
#define ITX_MODE_STAT 0
#define ITX_MODE_FILTER 1
#define ITX_ACCUM_DEFAULT 0
#define ITX_ACCUM_ATOMIC 1        /* global atomics straight from the classify kernel                  */
#define ITX_ACCUM_PARTITION 2     /* key emit -> radix partition -> LDS histograms (no scattered atomics) */

/* Record batch, structure of arrays. One element per BAM record, in file order:
 *   tid, pos            bam1_core_t.tid / .pos                       (cussamtools/bam.h:169-177)
 *   tmpend              n_cigar ? bam_calend(core, cigar) : pos + l_qseq          (generic.c:820)
 *   mapq                bam1_core_t.qual
 *   flag5               bit0 PAIRED(0x1) bit1 UNMAP(0x4) bit2 MUNMAP(0x8) bit3 REVERSE(0x10) bit4 READ1(0x40);
 *                       bit5 ITX_F5_NOLOOKUP, set by the caller: the record is counted like any other up to
 *                       reads_mapped / reads_mapped_unique (cnt[0..7], cnt[11]) but is not looked up — how a caller
 *                       applies the reference's two `continue`s that sit between those counters and the
 *                       accumulation: -R duplicates (generic.c:907-919) and the XA veto (generic.c:972-982)
 *   mpos, isize         bam1_core_t.mpos / .isize; both may be NULL when no record of the batch
 *                       has PAIRED set (then they are never read). */
typedef struct itx_batch {
    const int32_t *tid, *pos, *tmpend;
    const uint8_t *mapq, *flag5;
    const int32_t *mpos, *isize;
} itx_batch;
#define ITX_F5_NOLOOKUP 0x20
#define ITX_FLAG5(bamflag) ((uint8_t)((((bamflag) & 0x1) ? 1 : 0) | (((bamflag) & 0x4) ? 2 : 0) | (((bamflag) & 0x8) ? 4 : 0) | \
                                      (((bamflag) & 0x10) ? 8 : 0) | (((bamflag) & 0x40) ? 16 : 0)))

/* batch_capacity = largest n a single submit may carry. */
int itx_engine_create(const itx_table *t, const itx_params *p, size_t batch_capacity, itx_engine **out);
void itx_engine_destroy(itx_engine *e);

/* tid2chrom[tid] for the BAM header in use: index into chrom_size[], or -1 when the (possibly
 * -C renamed) reference name is not in the chrom-size file or its size is 2 (generic.c:793-801),
 * or -2 when -C drops it (generic.c:783-784). Call again when the next BAM of a list starts. */
int itx_engine_set_tidmap(itx_engine *e, const int32_t *tid2chrom, int n_tid);

/* Pinned double buffers (2 slots): the host decoder fills slot s while slot 1-s is in flight. */
typedef struct itx_staging {
    int32_t *tid, *pos, *tmpend;
    uint8_t *mapq, *flag5;
    int32_t *mpos, *isize;
    int32_t *hit_row;              /* filled by the device when submit_slot(want_hits != 0) */
    size_t capacity;
} itx_staging;
int itx_engine_staging(itx_engine *e, int slot, itx_staging *out);
/* Asynchronous: H2D copies + kernels on the slot's stream. has_paired = 0 promises that no record
 * has PAIRED set (mpos/isize are then not copied). want_hits: write back per record the index of
 * the chosen table row (as passed to itx_table_create) or -1 into staging.hit_row. */
int itx_engine_submit_slot(itx_engine *e, int slot, size_t n, int has_paired, int want_hits);
/* Same copies, classification only: nothing is accumulated, staging.hit_row receives the chosen rows. For
 * callers that must look at the chosen row before the record counts (the XA veto): classify, wait, set
 * ITX_F5_NOLOOKUP on the vetoed records, then submit the slot. */
int itx_engine_classify_slot(itx_engine *e, int slot, size_t n, int has_paired);
int itx_engine_wait_slot(itx_engine *e, int slot);

/* Same work on a batch that is ALREADY in device memory (pointers in `b` are device pointers, each
 * 16-byte aligned), enqueued on `stream` (a hipStream_t, NULL = the null stream). d_hit_row: optional
 * device int32[n] (16-byte aligned) for the chosen rows. Returns after enqueueing. Submissions of one
 * engine must be stream-ordered with respect to each other. */
int itx_engine_submit_device(itx_engine *e, const itx_batch *b, size_t n, int32_t *d_hit_row, void *stream);
/* The same on the engine's own compute stream (the one the slot submits run on: device batches and slot batches of one
 * record stream stay ordered). itx_engine_wait_own returns when everything submitted there is through — the arrays of
 * `b` may then be reused — without waiting for other work on the device. */
int itx_engine_submit_device_own(itx_engine *e, const itx_batch *b, size_t n, int32_t *d_hit_row);
int itx_engine_wait_own(itx_engine *e);
/* Classification only (no accumulation): cuskent/binRange.c:196-227 + generic.c:950-970 per record. */
int itx_engine_classify_device(itx_engine *e, const itx_batch *b, size_t n, int32_t *d_hit_row, void *stream);
/* The lookup alone, on plain intervals: for record i the FIRST row binKeeperFind(chrom of tid[i], pos[i], tmpend[i])
 * would return (cuskent/binRange.c:196-227, clipping included) — what cpgBedGraphOverlapRepeat takes for a CpG site
 * (generic.c:1082-1088) — or -1. mapq / flag5 are not looked at (the arrays must exist); no filter, no best-hit rule,
 * no -c, nothing is accumulated. Chosen rows come back like with classify (slot: staging.hit_row). */
int itx_engine_first_hit_slot(itx_engine *e, int slot, size_t n);
int itx_engine_first_hit_device(itx_engine *e, const itx_batch *b, size_t n, int32_t *d_hit_row, void *stream);
int itx_engine_sync(itx_engine *e);
/* Zero every accumulator. */
int itx_engine_reset(itx_engine *e);

/* What the loop leaves behind for the writers (generic.c:53-113, 1709-1746). Any pointer may be
 * NULL. cnt[13] as generic.c:1048-1060 (cnt[8] and cnt[12] stay 0 here: -R and the XA veto are
 * the caller's); rep/fam/cla_cnt: [0,n) all reads, [n,2n) reads with MAPQ >= mapq_min;
 * cov/cov_uniq: bp_total / bp_total_unique of every repName, concatenated per
 * itx_table_cov_offsets; locus_cnt[row]: slCount(ss->sl) per table row (filter mode). */
typedef struct itx_result {
    uint64_t *cnt;
    uint64_t *rep_cnt, *fam_cnt, *cla_cnt;
    uint32_t *cov, *cov_uniq;
    uint32_t *locus_cnt;
} itx_result;
/* Drains all streams, turns the raw accumulators into the result arrays on the device and copies
 * them to the host pointers of `out`. The raw accumulators are left untouched, so more batches may
 * follow and finish may be called again. */
int itx_engine_finish(itx_engine *e, const itx_result *out);

/* Multi-GPU: every output is a commutative integer sum over records, so ranks process disjoint
 * parts of the stream against replicated tables and exchange ONE compact partial at the end:
 *   itx_engine_partial_size   -> element counts of the partial: n_u64 x uint64 and n_u32 x uint32
 *   itx_engine_export_partial -> writes this engine's partial into caller-owned DEVICE buffers
 *                                (what the driver all-reduces with RCCL, sum, over xGMI)
 *   itx_engine_finish_partial -> like itx_engine_finish, but from (reduced) partial buffers
 * The partial is [cnt[16] | per-unit read counts] as u64 and [coverage difference arrays] (stat) or
 * [per-locus counts] (filter) as u32; sums are mod 2^64 / 2^32 like the reference's counters. */
int itx_engine_partial_size(const itx_engine *e, uint64_t *n_u64, uint64_t *n_u32);
int itx_engine_export_partial(itx_engine *e, void *d_u64, void *d_u32, void *stream);
int itx_engine_finish_partial(itx_engine *e, const void *d_u64, const void *d_u32, const itx_result *out);
/* The engine's own pair of partial buffers (device memory of the sizes above, owned by the engine): for a caller without
 * a device allocator of its own — export into them, reduce them across ranks (itx_comm_reduce_sum), finish from them. */
int itx_engine_partial_buffers(itx_engine *e, void **d_u64, void **d_u32);

/* The exchange itself, for a host that runs one process per GPU (iteres_amd/host: `iteres stat|filter` with ITX_GPUS /
 * ITX_RANK + ITX_WORLD): replaces nothing in the reference, which has one thread and one set of counters
 * (generic.c:705-722) — it is what makes N replicas of that loop one job. itx_comm_create: rank 0 publishes the
 * communicator id as the file `id_path`, the others wait for it. itx_comm_reduce_sum: sum over ranks of the two device
 * buffers of a partial (and of a small host vector `meta`, n_meta <= 64: the host's own counters) onto rank 0, in place,
 * behind `stream`'s work; returns when the result is there (other ranks: when their part is handed over).
 *   ITX_COMM_RCCL  ncclReduce over xGMI (librccl is loaded on first use)
 *   ITX_COMM_FILE  through files next to id_path, added on the host — for ranks that share a device (rehearsals on a
 *                  one-GPU box: RCCL refuses two ranks on one GPU)
 * Waiting is bounded (ITX_COMM_TIMEOUT seconds, default 900). */
typedef struct itx_comm itx_comm;
#define ITX_COMM_RCCL 0
#define ITX_COMM_FILE 1
int itx_comm_create(int rank, int world, int device, const char *id_path, int mode, itx_comm **out);
void itx_comm_destroy(itx_comm *c);
int itx_comm_reduce_sum(itx_comm *c, void *d_u64, size_t n64, void *d_u32, size_t n32, uint64_t *meta, size_t n_meta, void *stream);

/* Device time spent in the engine's kernels since the last reset, from HIP events recorded on the
 * submitting stream around each submit_device/submit_slot (milliseconds), and the number of
 * records classified. */
typedef struct itx_stats {
    double kernel_ms;             /* all kernels of all submits                                        */
    uint64_t records, hits;
    double stage_ms[5];           /* partition path, summed over submits: [0] stream (derive + classify + */
                                  /* key emit + partition count), [1] plan, [2] scatter, [3] hist, [4] 0  */
                                  /* — HIP events recorded between the launches on the submitting stream  */
    uint64_t submits, keys;       /* number of submits; keys emitted (partition path)                   */
} itx_stats;
int itx_engine_get_stats(itx_engine *e, itx_stats *out);

/* ---- BGZF inflate on the device --------------------------------------------------------------------------------
 * Replaces the reference's per-block zlib inflate (cussamtools/bgzf.c:367-397 inflate_block, reached from
 * bgzf_read -> bgzf_read_block, bgzf.c:425-521). The caller indexes the blocks of a chunk of the file (header
 * check bgzf.c:401-411, BSIZE, ISIZE trailer) and hands over the compressed chunk; every block is decoded by one
 * wavefront into its place in `out`. Offsets are relative to `comp` / `out`; uoff must be the running sum of the
 * usize's (the inflated bytes are contiguous). `comp` needs 16 readable bytes past comp_len. status[i] = 0 when block
 * i inflated to exactly usize bytes, else a small positive code (the data is damaged or the decoder declined it —
 * the caller's zlib has the last word, as in the reference). Synchronous; one calling thread per inflater. */
typedef struct itx_inflater itx_inflater;
#define ITX_BAMWIN_LANES 8        /* pushes that may be in flight at once (push_begin's s); ITX_PUSHES (environment) says how many an inflater sets up */
#define ITX_BAMWIN_LANES_DEFAULT 8  /* their kernels share ITX_LANES (default 4) compute lanes: the next push's bytes cross PCIe while a lane computes */
#define ITX_BAMWIN_WINDOWS 48     /* windows of inflated bytes (w): the decode may run this far ahead of the consumer (a window's
                                   * buffer is allocated when it is first pushed into: 288 GB of HBM is room for a long lead) */
typedef struct itx_bgzf_block {
    uint32_t coff, csize;         /* the whole gzip member: 18-byte header, deflate data, CRC32, ISIZE */
    uint32_t uoff, usize;
} itx_bgzf_block;
int itx_inflater_create(int device, itx_inflater **out);
void itx_inflater_destroy(itx_inflater *h);
int itx_inflate_bgzf(itx_inflater *h, const void *comp, size_t comp_len, const itx_bgzf_block *blk, size_t n_blk, void *out, size_t out_len,
                     uint8_t *status);
/* Device time of the two passes of the last call (Huffman -> tokens; first group of tokens -> bytes), milliseconds. */
int itx_inflater_last_ms(const itx_inflater *h, float *tokens_ms, float *resolve_ms);
/* ... and of pass 2 over ALL groups of that call; kernels only when the call was given out == NULL (the inflated bytes then
 * stay on the device: timing runs) */
int itx_inflater_last_resolve_all_ms(const itx_inflater *h, float *ms);

/* ---- BAM records located and parsed on the device ---------------------------------------------------------------
 * Replaces bam_read1 (cussamtools/bam.c:179-210) and the field reads of the scan loop (generic.c:745-905 off
 * bam1_core_t, bam.h:169-177; bam_calend bam.c:17-27) for whole chunks: the inflated bytes never leave the device,
 * only the per-record SoA (the itx_staging arrays) comes back. An inflater holds ITX_BAMWIN_WINDOWS WINDOWS of inflated bytes
 * (w) so that some can be filled while the records of another are still being fetched:
 *   push      inflate the blocks of a chunk into window w (offsets as for itx_inflate_bgzf; *n_new = bytes added)
 *   patch     overwrite part of what push produced (a block the caller inflated itself); truncate: drop the end
 *   carry     move the unconsumed tail of window `from` (a partial record) in front of window `to`'s fresh bytes
 *   peek/skip the unconsumed bytes from the front (the BAM header is parsed by the caller)
 *   parse     locate every complete record of window w and parse it: *n_rec records; *malformed = a record length
 *             below 32 ended the stream (bam.c:186-190); *flags bit 0: some record has PAIRED set, bit 1: some
 *             record carries an XA tag; *rewalked: pieces whose guessed start had to be corrected (may be NULL)
 *   fetch     records [first, first + n) of the last parse into dst's HOST arrays at index dst_at (hit_row untouched),
 *             optionally their byte offsets in the window and per-record XA marks
 *   bytes     raw window bytes at such offsets (read names, XA / NM strings: the caller's side channels)
 *   tids      which references (tid < n_targets) have a mapped record in the last parsed window (one byte each)
 *   device_batch  records [first, ..) of the last parse as DEVICE arrays for itx_engine_submit_device* (first % 16 == 0;
 *             valid until the next parse)
 * Records are located by guess-and-verify (csrc/itx_inflate.hip): exact whatever the bytes look like. A record of
 * more than 4 MiB that straddles two chunks makes carry move the fresh bytes back (ITX_E_LIMIT only when the window's buffer
 * cannot hold both). One thread may push while another
 * parses / fetches the OTHER window. */
/* Before a stream of pushes, while the device is idle: every buffer the pushes will need, sized for chunks of at most
 * comp_bytes compressed bytes / max_blocks blocks / max_bytes inflated bytes, so that nothing is allocated (and, worse,
 * freed: hipFree waits for every kernel in flight) inside the pipeline — ONE device allocation for all of it. *n_windows,
 * in: the most the caller can use (< 1: no preference); out: how many windows were set up (w < *n_windows; between 4 and
 * ITX_BAMWIN_WINDOWS, by what the device has free). Optional: without it the buffers grow on demand. */
int itx_inflater_reserve(itx_inflater *h, size_t comp_bytes, size_t max_blocks, size_t max_bytes, int *n_windows);
int itx_bamwin_push(itx_inflater *h, int w, const void *comp, size_t comp_len, const itx_bgzf_block *blk, size_t n_blk, uint8_t *status, size_t *n_new);
/* push in two halves, for a caller that keeps several pushes going (s < ITX_BAMWIN_LANES: each has its own stream and scratch,
 * so the latency-bound Huffman pass of one chunk runs beside the replay of the previous ones): begin enqueues and returns;
 * copied waits until `comp` may be reused; end waits for the push and delivers status / n_new. */
int itx_bamwin_push_begin(itx_inflater *h, int w, int s, const void *comp, size_t comp_len, const itx_bgzf_block *blk, size_t n_blk);
int itx_bamwin_push_copied(itx_inflater *h, int s);
int itx_bamwin_push_end(itx_inflater *h, int s, uint8_t *status, size_t *n_new);
int itx_bamwin_patch(itx_inflater *h, int w, size_t uoff, const void *bytes, size_t len);
int itx_bamwin_truncate(itx_inflater *h, int w, size_t n_new);
int itx_bamwin_carry(itx_inflater *h, int from, int to);
int itx_bamwin_avail(const itx_inflater *h, int w, size_t *bytes);
int itx_bamwin_peek(itx_inflater *h, int w, size_t off, void *dst, size_t len);
int itx_bamwin_skip(itx_inflater *h, int w, size_t n);
int itx_bamwin_parse(itx_inflater *h, int w, int n_targets, size_t *n_rec, int *malformed, int *flags, size_t *rewalked);
int itx_bamwin_fetch(itx_inflater *h, size_t first, size_t n, const itx_staging *dst, size_t dst_at, uint32_t *rec_off, uint8_t *xa);
int itx_bamwin_bytes(itx_inflater *h, size_t off, void *dst, size_t len);
int itx_bamwin_tids(itx_inflater *h, uint8_t *seen, int n_targets);
int itx_bamwin_device_batch(itx_inflater *h, size_t first, int with_mates, itx_batch *out);

/* ---- the XA / NM multi-mapping veto on the device --------------------------------------------------------------------
 * Replaces mapped2diffSubfam (generic.c:303-341) and its gate (generic.c:972-982) for the records of a window the device
 * decoder has parsed: the XA:Z / NM tags are read where the inflated record lies, the alternatives with NM' <= NM are looked
 * up in the table ("any overlapping row of another subfamily name", names compared without case), and the records the
 * reference would drop get ITX_F5_NOLOOKUP in the window's flag5 array — before the window is handed to the engine.
 *   create       row_rep[n_rows]: repName id of the caller's row i; rep_word[n_rep]: equal for names that sameWord() calls
 *                equal; chrom_name[n_chrom]: the chromosome names in chrom_size[] order (hashRmsk's keys); p: the run's
 *                parameters (the record's interval is re-derived for qlen, generic.c:764-905)
 *   set_tidmap   per BAM header, as itx_engine_set_tidmap
 *   hits/stream  where itx_engine_classify_device must leave the chosen rows of the batch, and on which stream
 *   itx_bamwin_xa_veto  judges records [first, first + n) of the inflater's last parsed window (n <= batch_capacity):
 *                *n_hard > 0: some record needs the host's reading (a number strtol(.., 0, 0) reads differently from plain
 *                decimal, an alternative without four fields — where the reference asserts): NOTHING was marked, the caller
 *                takes the host route for these records; else *n_vetoed records were marked. */
typedef struct itx_xaveto itx_xaveto;
int itx_xaveto_create(const itx_table *t, const itx_params *p, const uint32_t *row_rep, const uint32_t *rep_word, const char *const *chrom_name, int n_chrom,
                      size_t batch_capacity, itx_xaveto **out);
void itx_xaveto_destroy(itx_xaveto *x);
int itx_xaveto_set_tidmap(itx_xaveto *x, const int32_t *tid2chrom, int n_tid);
int32_t *itx_xaveto_hits(itx_xaveto *x);
void *itx_xaveto_stream(itx_xaveto *x);
int itx_bamwin_xa_veto(itx_inflater *h, itx_xaveto *x, size_t first, size_t n, uint64_t *n_vetoed, uint64_t *n_hard);

/* ---- a backlog of parsed records in HBM -----------------------------------------------------------------------------
 * What a caller keeps of windows it parses BEFORE its table exists (the engine cannot take them yet): the per-record arrays
 * only — 14 B per record, 22 with mates, a ninth of the inflated bytes — so that the windows can be pushed into again and the
 * decode goes on while the rmsk file is parsed and the table built (host/stream.c: the helper thread). append copies n records
 * (device arrays) and says where they lie (a multiple of 16); batch gives that place back as an itx_batch. */
typedef struct itx_backlog itx_backlog;
int itx_backlog_create(int device, size_t max_records, itx_backlog **out);
void itx_backlog_destroy(itx_backlog *b);
size_t itx_backlog_room(const itx_backlog *b);
int itx_backlog_append(itx_backlog *b, const itx_batch *src, size_t n, size_t *at);
int itx_backlog_batch(const itx_backlog *b, size_t at, int with_mates, itx_batch *out);

/* ---- -R (remove redundant reads) on the device --------------------------------------------------------------------
 * Replaces generic.c:907-919 (filter copy 544-556): the `dup` hash of "chr:start:end:strand" keys and the `continue` behind
 * it. As a rule per record, records numbered in file order: one with MAPQ >= -Q is dropped iff an earlier one with MAPQ >= -Q
 * has the same (chromosome name, start, end, strand); one with MAPQ < -Q looks up the key of the last MAPQ >= -Q record before
 * it and is always dropped — except the very first record to reach that point of the loop when it comes before any MAPQ >= -Q
 * record (the reference's key buffer then holds no record's key: modelled as a key no record has). Dropped records get
 * ITX_F5_NOLOOKUP in their flag5 (the engine counts them up to reads_mapped(_unique) and does not look them up); the caller
 * corrects cnt[11] by *dup_unique. csrc/itx_dedup.hip: a hash table in HBM that only ever compares FULL keys (hashes pick
 * cells), grown by rehashing; at most 2^32 - 1 records per run.
 *   create     chrom_size[n_chrom] as for itx_table_create; p: mapq_min, extension, isize_max, treat_pe_as_se,
 *              discard_half_mapped; first_cells: a first size for the table (it grows)
 *   set_tidmap the BAM header in use: tid2chrom as for itx_engine_set_tidmap, tid2name[t] an id of the (renamed) reference
 *              name that is equal for equal strings over all files of the run (< 2^31)
 *   run        the next n records of the stream as DEVICE arrays (mpos / isize NULL: no record is paired), in file order,
 *              window after window; synchronous
 *   itx_bamwin_dedup   run over all records of the inflater's last parsed window */
typedef struct itx_dedup itx_dedup;
int itx_dedup_create(int device, const int64_t *chrom_size, int n_chrom, const itx_params *p, size_t first_cells, itx_dedup **out);
void itx_dedup_destroy(itx_dedup *d);
int itx_dedup_set_tidmap(itx_dedup *d, const int32_t *tid2chrom, const uint32_t *tid2name, int n_tid);
int itx_dedup_run(itx_dedup *d, const int32_t *tid, const int32_t *pos, const int32_t *tmpend, const uint8_t *mapq, uint8_t *flag5, const int32_t *mpos,
                  const int32_t *isize, size_t n);
int itx_dedup_counts(itx_dedup *d, uint64_t *dup_unique, uint64_t *dropped, uint64_t *keys);
int itx_bamwin_dedup(itx_inflater *h, itx_dedup *d);

/* ITX_TIMING: what the device decoder measured about itself (pushes, mean duration of the two passes, device allocations),
 * one line on stderr; also printed when the process exits normally. */
void itx_timing_report(void);

/* Page-locked host memory for the buffers that cross PCIe on every call (NULL when it cannot be had). */
void *itx_pinned_alloc(size_t bytes);
void itx_pinned_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
