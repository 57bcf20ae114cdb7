// itx_common.h — internal definitions shared by the table builder, the kernels and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/iteres_amd.h"

#define ITX_LOGW 13         // slots per partition / LDS window of the partition path (W = 8192)

// One interval of the device table, 32 bytes (two dwordx4 loads). Intervals of a chromosome are
// contiguous and sorted by (start, file order).
struct __attribute__((aligned(16))) ItxIv {
    // first 16 bytes: all the overlap scan reads
    int32_t  s, e;        // genomic [s, e)
    int32_t  pbelow;      // max e over this chromosome's intervals BELOW this one (INT32_MIN for the first): a scan that
                          // walks down from here goes on only while pbelow > query start
    uint32_t rank;        // position in binKeeperFind's list order within the chromosome (binRange.c:209-225)
    // second 16 bytes: read once, for the chosen row
    uint32_t cs;          // consensus_start as the reference parses it (generic.c:1596-1600)
    uint32_t jcap;        // min(consensus_end, repeat length): first consensus index NOT incremented
    uint32_t covslot;     // first slot of the interval's unit inside the slot space (rep_len+1 slots per unit)
    uint32_t zslot;       // covslot + rep_len: the unit's extra slot (never part of the coverage output)
};
static_assert(sizeof(ItxIv) == 32, "ItxIv must be 32 bytes");

// Device view of the table (passed to kernels by value).
struct ItxDevTable {
    const ItxIv    *iv;         // [n_rows]
    const int32_t  *orig;       // [n_rows] sorted index -> caller's row index
    const uint32_t *row_unit;   // [n_rows] sorted index -> unit (ITX_ACCUM_ATOMIC counts reads per unit directly)
    const uint32_t *unit_slot;  // [n_units + 1] first slot of every unit
    const uint32_t *part_unit;  // [n_slots >> ITX_LOGW, + 2] first unit that reaches into each window of 2^ITX_LOGW slots
    const uint2    *bl;         // binned index, one slice per chromosome (ItxTidRec.bin_base), per bin b of 2^shift bp:
                                //   .x = first index with s >= (b << shift)          (upper bound of "s < x" queries)
                                //   .y = first index whose prefix-max end > (b << shift)      (lower bound of what can still overlap)
    int32_t  shift;
    uint32_t n_rows;
    uint32_t n_units;
    uint32_t n_slots;           // sum over units of (rep_len + 1)
};

// Everything the kernels need to know about one BAM reference id, 32 bytes: one load per record
// (wave-uniform for coordinate-sorted input) instead of a chain of per-chromosome lookups.
struct __attribute__((aligned(16))) ItxTidRec {
    int32_t  chrom;      // index into chrom_size[], -1 unknown / size 2 (generic.c:793-801), -2 dropped by -C (generic.c:783-784)
    int32_t  size;       // chromosome size (int, as binKeeperNew(0, size) holds it)
    uint32_t iv_lo, iv_hi;   // the chromosome's interval range in the table
    uint32_t bin_base;   // start of the chromosome's slice of ItxDevTable.bl
    uint32_t pad[3];
};
static_assert(sizeof(ItxTidRec) == 32, "ItxTidRec must be 32 bytes");

// Raw device accumulators (element offsets).
//   u64: cnt[16] | unit_all[n_units] | unit_uniq[n_units]          reads that chose a row of the unit (all / MAPQ >= -Q)
//   u32: D_all[n_slots] | D_uniq[n_slots] | locus[n_rows]          D = starts - ends of consensus ranges, per slot
// A unit's coverage is the prefix sum of D over its slots (mod 2^32 like the reference's unsigned counters); read counts
// of names, families and classes are sums of unit counts. The u64 block and the two D arrays ARE the partial a
// multi-GPU driver reduces (include/iteres_amd.h: itx_engine_export_partial).
struct ItxAccumLayout {
    uint64_t d_all, d_uniq, locus, n_u32;
    uint64_t n_units;          // u64: unit_all at 16, unit_uniq at 16 + n_units
};
static inline ItxAccumLayout itx_accum_layout(uint64_t n_slots, uint64_t n_rows, uint64_t n_units)
{
    ItxAccumLayout L;
    const uint64_t st = (n_slots + 63) & ~uint64_t(63);      // each array starts 256-byte aligned (16-byte vector updates)
    L.d_all = 0; L.d_uniq = st; L.locus = 2 * st;
    L.n_u32 = L.locus + n_rows;
    L.n_units = n_units;
    return L;
}
// The partial a multi-GPU driver reduces (include/iteres_amd.h: itx_engine_export_partial):
//   u64: cnt[16] | unit_all[n_units] | unit_uniq[n_units]
//   u32: stat  : D_all[n_slots] | D_uniq[n_slots]   (D = A - B)
//        filter: locus[n_rows]  (per SORTED row)

struct itx_table {
    int device;
    int n_chrom, shift;
    uint32_t n_rows, n_rep, n_fam, n_cla, n_units, n_slots;
    uint64_t cov_len;
    uint64_t table_bytes;
    ItxDevTable dev;
    // host copies used by finish()/info/set_tidmap
    uint32_t *h_rep_len;     // [n_rep]
    uint32_t *h_chrom_off;   // [n_chrom+1]
    uint32_t *h_bin_off;     // [n_chrom+1]
    int32_t  *h_chrom_size;  // [n_chrom]
    void *d_all;             // single device allocation backing every table array
    // per unit (device): slot range, ids, where its coverage lands, whether it is its name's only unit
    uint32_t *d_unit_slot;   // [n_units+1]
    uint4    *d_unit_ids;    // [n_units] (rep, fam, cla, solo)
    uint64_t *d_unit_covoff; // [n_units] offset of the unit's repName inside the concatenated coverage vectors
    uint32_t *d_row_unit;    // [n_rows] (sorted order)
    uint32_t *d_part_unit;
};

void itx_set_error(const char *fmt, ...);
#define ITX_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t err__ = (call);                                                            \
        if (err__ != hipSuccess) {                                                            \
            itx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__, __LINE__); \
            return ITX_E_NO_DEVICE;                                                           \
        }                                                                                     \
    } while (0)

// Per-record parameters handed to the kernels.
struct ItxRunParams {
    uint32_t mapq_min;
    float    min_cov;
    float    cov_hi, cov_lo;     // overlap >= qlen * cov_hi: certainly passes -c; overlap < qlen * cov_lo: certainly fails (itx_cov_bounds)
    uint32_t extension;
    uint32_t isize_max;
    int32_t  treat, discard, mode;
    int32_t  n_tid;
    const ItxTidRec *tidrec;    // device, [n_tid]
};

// The -c test of the reference is `fl32(overlap / qlen) < min_cov` (generic.c:296-301, 961-962). For overlaps and
// lengths below 2^23 the kernels decide it with one multiply: overlap >= fl32(qlen * hi) implies the quotient
// rounds to >= min_cov, overlap < fl32(qlen * lo) implies it rounds below; hi / lo sit 16 ulps either side of
// min_cov (relative 2^-20, far beyond the 2^-24 roundings involved). Whatever falls between, and every min_cov
// that is not a comfortably normal positive number, takes the exact division.
static inline void itx_cov_bounds(float m, float *hi, float *lo)
{
    if (!(m > 0.0f)) {                      // zero, negative, NaN: `c < m` is never true, every hit passes
        *hi = -__builtin_inff();
        *lo = -__builtin_inff();
        return;
    }
    if (__builtin_isinf(m)) {               // `c < inf` always true
        *hi = __builtin_inff();
        *lo = __builtin_inff();
        return;
    }
    if (m < 1e-30f || m > 1e30f) {          // near the ends of the exponent range: always divide
        *hi = __builtin_inff();
        *lo = -__builtin_inff();
        return;
    }
    float h = m, l = m;
    for (int i = 0; i < 16; i++) {
        h = __builtin_nextafterf(h, __builtin_inff());
        l = __builtin_nextafterf(l, 0.0f);
    }
    *hi = h;
    *lo = l;
}

struct ItxDevBatch {
    const int32_t *tid, *pos, *tmpend;
    const uint8_t *mapq, *flag5;
    const int32_t *mpos, *isize;   // may be null
};

// Streaming kernel (itx_stream.hip): one launch classifies n records and, per `what`, does nothing else,
// accumulates with global atomics (stat: A/B arrays, filter: per-locus counts) or emits keys.
enum { ITX_DO_CLASSIFY = 0, ITX_DO_ATOMIC_STAT = 1, ITX_DO_ATOMIC_LOCUS = 2, ITX_DO_EMIT = 3,
       ITX_DO_FIND_FIRST = 4 /* plain intervals (pos, tmpend): first overlapping row in binKeeperFind's order, nothing else */,
       ITX_DO_EMIT_WIDE = 5 /* EMIT with partitions of more than 2^ITX_LOGW slots: chosen by itx_launch_stream from the plan's log_w */ };
#define ITX_STREAM_LB 5             // workgroups of the streaming kernel per CU its register budget is set for (96 VGPRs)
// Workgroups of a streaming launch over n records: whole ROUNDS of what the chip holds at once (CUs x ITX_STREAM_LB), each
// workgroup walking one contiguous region of about ITX_STREAM_REGION records. One round of long regions ends with the chip
// half empty while the slowest regions finish (measured at 500 M records: 1 round 2.96 ms, 12 rounds 2.71 ms, and the
// scatter that follows 1.36 -> 1.03 ms); regions much shorter than this cost the partition path more in bookkeeping than
// the balance gains (50 M records: best at one round = 39 k records each). ITX_STREAM_BLOCKS / _REGION / _ROUNDS override.
#define ITX_STREAM_REGION 32768u
#define ITX_STREAM_ROUNDS 16u
unsigned itx_stream_blocks(int device, size_t n);
// span of records per workgroup (a multiple of the stream tile) and the workgroups that cover n records with it
void itx_stream_plan(unsigned blocks, size_t n, size_t *span, unsigned *n_blocks);
#define ITX_STREAM_TILE 1024u      // records per workgroup iteration (4 waves x 64 lanes x 4 records)
#define ITX_PART_SUB 8u            // sub-cursors per partition (partition path)
// What the emitting launch needs to know about the partition path's bookkeeping.
struct ItxEmitPlan {
    uint32_t *subcur;   // [n_part * ITX_PART_SUB] zeroed before the launch: keys reserved per (partition, sub-cursor)
    uint32_t *offm;     // [n_blocks][n_part] offset of each workgroup's keys inside (partition, sub-cursor)
    uint32_t n_part;
    uint32_t log_w;     // log2(slots per partition)
};
int itx_launch_stream(int what, const ItxDevTable &T, const ItxRunParams &P, const ItxDevBatch &B, size_t n, size_t span,
                      unsigned n_blocks, int32_t *d_hit_row, uint64_t *u64, uint32_t *u32, const ItxAccumLayout &L, uint2 *keys0,
                      uint32_t *blk_cnt, const ItxEmitPlan &E, hipStream_t st);
// finish-time kernels (itx_finalize.hip)
int itx_launch_export(const itx_table *t, int mode, const uint64_t *u64, const uint32_t *u32, const ItxAccumLayout &L, uint64_t *p64,
                      uint32_t *p32, hipStream_t st);
int itx_launch_finish(const itx_table *t, int mode, const uint64_t *p64, const uint32_t *p32, uint64_t *d_counts /*2*(rep+fam+cla)*/,
                      uint32_t *d_cov, uint32_t *d_cov_uniq, uint32_t *d_locus_out, hipStream_t st);
