// itx_common.h — internal definitions shared by the table builder, the kernels and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/iteres_amd.h"

// One interval of the device table, 32 bytes (two dwordx4 loads). Intervals of a chromosome are
// contiguous and sorted by (start, file order).
struct __attribute__((aligned(16))) ItxIv {
    int32_t  s, e;        // genomic [s, e)
    int32_t  pmax_e;      // max e over this chromosome's intervals [first .. this]: scan-stop bound
    uint32_t cs;          // consensus_start as the reference parses it (generic.c:1596-1600)
    uint32_t jcap;        // min(consensus_end, repeat length): first consensus index NOT incremented
    uint32_t covslot;     // first slot of the interval's repName inside the slot space (rep_len+1 slots per name)
    uint32_t zslot;       // covslot + rep_len: the name's extra slot (never part of the coverage output)
    uint32_t famcla;      // fam << 16 | cla
};
static_assert(sizeof(ItxIv) == 32, "ItxIv must be 32 bytes");

// Device view of the table (passed to kernels by value).
struct ItxDevTable {
    const ItxIv    *iv;         // [n_rows]
    const uint32_t *rank;       // [n_rows] position in binKeeperFind's list order within the chromosome
    const int32_t  *orig;       // [n_rows] sorted index -> caller's row index
    const uint32_t *chrom_off;  // [n_chrom+1] interval range of each chromosome
    const uint32_t *bin_off;    // [n_chrom+1] start of each chromosome's slice of bidx
    const uint32_t *bidx;       // bidx[bin_off[c] + b] = chrom_off[c] + #{intervals of c with s < (b << shift)}
    const int32_t  *chrom_size; // [n_chrom] (int, as binKeeperNew(0, size) holds it)
    int32_t  n_chrom;
    int32_t  shift;
    uint32_t n_rows;
    uint32_t n_rep, n_fam, n_cla;
    uint32_t n_slots;           // sum(rep_len + 1)
};

// Layout of the accumulator blocks (element offsets).
//   u64 block: cnt[16] | rep[2*n_rep] | fam[2*n_fam] | cla[2*n_cla]
//   u32 block: A_all[n_slots] | A_uniq[n_slots] | B_all[n_slots] | B_uniq[n_slots] | locus[n_rows]
// A = "range starts here" counts, B = "range ends here" counts in slot space; coverage of a name is the
// prefix sum of A-B over its slots, its read count is sum(A) (every classified read contributes exactly
// one start). All sums are mod 2^32 / 2^64 like the reference's unsigned counters.
struct ItxAccumLayout {
    uint64_t cnt, rep, fam, cla, n_u64;
    uint64_t a_all, a_uniq, b_all, b_uniq, locus, n_u32;
};
static inline ItxAccumLayout itx_accum_layout(uint64_t n_rep, uint64_t n_fam, uint64_t n_cla, uint64_t n_slots, uint64_t n_rows)
{
    ItxAccumLayout L;
    L.cnt = 0; L.rep = 16; L.fam = L.rep + 2 * n_rep; L.cla = L.fam + 2 * n_fam; L.n_u64 = L.cla + 2 * n_cla;
    L.a_all = 0; L.a_uniq = n_slots; L.b_all = 2 * n_slots; L.b_uniq = 3 * n_slots; L.locus = 4 * n_slots;
    L.n_u32 = L.locus + n_rows;
    return L;
}

struct itx_table {
    int device;
    int n_chrom, shift;
    uint32_t n_rows, n_rep, n_fam, n_cla, n_slots;
    uint64_t cov_len;
    uint64_t table_bytes;
    ItxDevTable dev;
    // host copies used by finish()/info
    uint32_t *h_rep_len;     // [n_rep]
    uint32_t *h_covslot;     // [n_rep+1] first slot of each name
    void *d_all;             // single device allocation backing every table array
    uint32_t *d_rep_len;     // [n_rep]
    uint32_t *d_covslot;     // [n_rep+1]
    uint64_t *d_covoff;      // [n_rep+1] offsets into the concatenated coverage vectors
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
    uint32_t extension;
    uint32_t isize_max;
    int32_t  treat, discard, mode;
    int32_t  n_tid;
    const int32_t *tid2chrom;   // device
};

struct ItxDevBatch {
    const int32_t *tid, *pos, *tmpend;
    const uint8_t *mapq, *flag5;
    const int32_t *mpos, *isize;   // may be null
};

// kernels.hip launch wrappers -----------------------------------------------------------------
struct ItxWork;   // scratch owned by the engine (partition path)
int itx_launch_atomic(const ItxDevTable &T, const ItxRunParams &P, const ItxDevBatch &B, size_t n, int do_accum,
                      int32_t *d_hit_row, uint64_t *u64, uint32_t *u32, const ItxAccumLayout &L, hipStream_t st);
int itx_launch_finalize(const itx_table *t, const uint64_t *u64, const uint32_t *u32, const ItxAccumLayout &L,
                        uint64_t *d_rep_out, uint32_t *d_cov, uint32_t *d_cov_uniq, hipStream_t st);
