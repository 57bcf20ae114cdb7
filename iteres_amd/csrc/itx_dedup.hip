// itx_dedup.hip — `-R` (remove redundant reads) on the device.
//
// What it replaces: generic.c:907-919 (stat copy; filter copy 544-556) —
//     if (MAPQ >= Q) sprintf(key, "%s:%u:%u:%c", chr, start, end, strand);      // otherwise `key` keeps what it held
//     if (hashLookup(dup, key) == NULL) hashAddInt(dup, key, 1); else continue;
// for every record that has reached that point of the loop (mapped, reference usable, pair rules passed: the records the
// reference counts as reads_mapped). Written down as a rule per record, with the records numbered in file order:
//   * a record with MAPQ >= Q is dropped iff an EARLIER record with MAPQ >= Q has the same (chromosome name, start, end, strand);
//   * a record with MAPQ < Q looks up the key of the last MAPQ >= Q record before it — which is in the set by then — and is
//     always dropped; before the first MAPQ >= Q record the key buffer holds no record's key (the reference reads its
//     uninitialised stack there; modelled, like the host route does, as one key no record can produce): the first record of
//     all inserts it and is kept, the others find it.
// So: kept(i) = uniq(i) ? no earlier uniq record with the same key : i is the first record to reach this point at all.
//
// Device form. A window of records (they come in file order, window after window; inside a window the threads run in any order):
//   k_dd_claim   every uniq record finds or claims the cell of its key's 64-bit HASH (one compare-and-swap on the hash word: the
//                cell's identity is that word, so nobody ever has to read a key another thread is still writing) and takes the
//                cell's record number down to its own (atomicMin): afterwards a cell holds the FIRST record with that hash;
//                all records that reach the point also take `first_ok` down.
//   k_dd_owner   the record a cell's number names writes its full key into the cell (one writer per cell).
//   k_dd_verdict every uniq record compares its full key with its cell's: equal and the cell's number smaller -> duplicate
//                (ITX_F5_NOLOOKUP into the window's flag array, the count for cnt[11]); different -> two keys share a hash
//                (about n^2 / 2^65 of the runs: 1 % at 5e8 records) — such a record goes to a small overflow list that a
//                single workgroup settles exactly, by full key, in record order (k_dd_overflow). Records with MAPQ < Q are
//                dropped unless their number is `first_ok`.
// The table grows by rehashing on the device (cells never move otherwise). Everything is exact: hashes only pick cells.
#include "itx_device.h"

#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define DD_HIP(call)                                                                                     \
    do {                                                                                                  \
        hipError_t err__ = (call);                                                                        \
        if (err__ != hipSuccess) {                                                                        \
            itx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__, __LINE__); \
            return ITX_E_NO_DEVICE;                                                                       \
        }                                                                                                 \
    } while (0)

#define DD_EMPTY 0xffffffffffffffffull
#define DD_NOIDX 0xffffffffu
#define DD_OVER_CAP 65536u                 // overflow entries (records whose key shares a 64-bit hash with another key) the list holds

struct DdParams {
    uint32_t mapq_min, extension, isize_max;
    int32_t treat, discard;
    const int2 *tid;                       // [n_tid]: (chromosome index or < 0, chromosome size)
    const uint32_t *tid_name;              // [n_tid]: identity of the chromosome's (renamed) name, equal across files for equal strings
    int32_t n_tid;
    unsigned long long hash_mask;          // all ones; fewer bits (ITX_DEDUP_HASH_BITS, tests) make keys share hashes, which the overflow list has to settle
};

struct DdTable {
    unsigned long long *h;                 // [cap] the key's hash, DD_EMPTY: free
    uint32_t *idx;                         // [cap] first record (number in file order) with that hash
    unsigned long long *ka;                // [cap] start | end << 32
    uint32_t *kb;                          // [cap] name id << 1 | strand
    uint32_t mask;
};

struct DdOver {                            // a record whose key differs from the key its hash's cell holds
    unsigned long long ka;
    uint32_t kb, idx;
};

struct DdState {
    uint32_t first_ok;                     // smallest number of a record that reached the point (DD_NOIDX: none yet)
    uint32_t n_cells;                      // cells in use
    uint32_t n_over;                       // entries of the overflow list
    uint32_t pad;
    unsigned long long dup_unique, dropped;
};

// generic.c:764-905 for one record (iteres_amd/host/side.c host_derive, itx_stream.hip derive_one): does it reach the point, with which key
static __device__ inline bool dd_key(const DdParams &P, int32_t t, int32_t pos, int32_t tmpend, uint32_t mq, uint32_t f5, int32_t mpos, int32_t isz,
                                     unsigned long long *ka, uint32_t *kb, bool *uniq)
{
    if (f5 & F5_UNMAP) return false;                                   // generic.c:764
    const int2 tr = (t >= 0 && t < P.n_tid) ? P.tid[t] : make_int2(-1, 0);
    if (tr.x < 0) return false;                                        // generic.c:781-801
    const uint32_t cend = (uint32_t)(tr.y - 1);                        // generic.c:796
    if (cend == 1u) return false;
    bool se;
    if (P.treat || !(f5 & F5_PAIRED)) {
        se = true;
    } else if (!(f5 & F5_MUNMAP)) {                                    // generic.c:836-860
        if (!(f5 & F5_READ1)) return false;
        const uint32_t a = isz < 0 ? 0u - (uint32_t)isz : (uint32_t)isz;
        if (a > P.isize_max || isz == 0) return false;
        se = false;
    } else {
        if (P.discard) return false;                                   // generic.c:862-863
        se = true;
    }
    uint32_t st, en, strand;
    if (se) {                                                          // generic.c:819-833
        st = (uint32_t)pos;
        en = cend < (uint32_t)tmpend ? cend : (uint32_t)tmpend;
        strand = (f5 & F5_REVERSE) ? 1u : 0u;
        if (P.extension) {
            if (!strand) {
                const uint32_t e2 = st + P.extension;
                en = e2 < cend ? e2 : cend;
            } else {
                st = en < P.extension ? 0u : en - P.extension;
            }
        }
    } else if (isz > 0) {                                              // generic.c:845-855
        st = (uint32_t)pos;
        const uint32_t e2 = st + (uint32_t)isz;
        en = cend < e2 ? cend : e2;
        strand = 0u;
    } else {
        st = (uint32_t)mpos;
        const uint32_t e2 = st - (uint32_t)isz;
        en = cend < e2 ? cend : e2;
        strand = 1u;
    }
    *ka = (unsigned long long)st | (unsigned long long)en << 32;
    *kb = P.tid_name[t] << 1 | strand;
    *uniq = mq >= P.mapq_min;
    return true;
}

static __device__ __host__ inline unsigned long long dd_hash(unsigned long long ka, uint32_t kb)
{
    unsigned long long x = ka * 0x9e3779b97f4a7c15ull;
    x ^= x >> 32;
    x += (unsigned long long)kb * 0xc2b2ae3d27d4eb4full;
    x ^= x >> 29;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 32;
    return x == DD_EMPTY ? 0x1234567ull : x;
}

// the cell whose hash word is `hv`, claiming a free one when there is none yet (insert = true; *fresh says so) or DD_NOIDX (insert = false)
static __device__ inline uint32_t dd_cell(const DdTable &T, unsigned long long hv, bool insert, bool *fresh)
{
    uint32_t j = (uint32_t)((hv * 0x9e3779b97f4a7c15ull) >> 32) & T.mask;
    for (;;) {
        unsigned long long cur = __hip_atomic_load(&T.h[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == hv) return j;
        if (cur == DD_EMPTY) {
            if (!insert) return DD_NOIDX;
            const unsigned long long old = atomicCAS(&T.h[j], DD_EMPTY, hv);
            if (old == DD_EMPTY) {
                *fresh = true;
                return j;
            }
            if (old == hv) return j;
        }
        j = (j + 1) & T.mask;                                          // the load factor stays below 0.7: a free cell always comes
    }
}

// one atomic per wave for a per-lane 0/1 (the counters are single words: a window's millions of records would queue up on them)
static __device__ inline void dd_count(unsigned long long *ctr, bool mine)
{
    const unsigned long long m = __ballot(mine);
    if (m && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)m) - 1u) atomicAdd(ctr, (unsigned long long)__popcll(m));
}
static __device__ inline void dd_count32(uint32_t *ctr, bool mine)
{
    const unsigned long long m = __ballot(mine);
    if (m && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)m) - 1u) atomicAdd(ctr, (uint32_t)__popcll(m));
}

#define DD_LOAD(i)                                                                                                                      \
    const int32_t t = tid_a[i], pos = pos_a[i], tmpend = end_a[i], mpos = mpos_a ? mpos_a[i] : 0, isz = isize_a ? isize_a[i] : 0; \
    const uint32_t mq = mapq_a[i], f5 = f5_a[i];                                                                                    \
    unsigned long long ka = 0;                                                                                                      \
    uint32_t kb = 0;                                                                                                                \
    bool uniq = false;                                                                                                              \
    const bool ok = dd_key(P, t, pos, tmpend, mq, f5, mpos, isz, &ka, &kb, &uniq)

#define DD_ARGS                                                                                                                          \
    const int32_t *__restrict__ tid_a, const int32_t *__restrict__ pos_a, const int32_t *__restrict__ end_a, const uint8_t *__restrict__ mapq_a, \
        const int32_t *__restrict__ mpos_a, const int32_t *__restrict__ isize_a, uint32_t n, uint32_t base

__global__ __launch_bounds__(256) void k_dd_claim(DdParams P, DdTable T, DdState *S, const uint8_t *__restrict__ f5_a, DD_ARGS)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    bool fresh = false;
    if (i < n) {
        DD_LOAD(i);
        if (ok) {
            const uint32_t me = base + i;
            if (me < __hip_atomic_load(&S->first_ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&S->first_ok, me);       // (false for all but a window's first few)
            if (uniq) {
                const uint32_t j = dd_cell(T, dd_hash(ka, kb) & P.hash_mask, true, &fresh);
                atomicMin(&T.idx[j], me);
            }
        }
    }
    dd_count32(&S->n_cells, fresh);
}

__global__ __launch_bounds__(256) void k_dd_owner(DdParams P, DdTable T, const uint8_t *__restrict__ f5_a, DD_ARGS)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    DD_LOAD(i);
    if (!ok || !uniq) return;
    bool fr = false;
    const uint32_t j = dd_cell(T, dd_hash(ka, kb) & P.hash_mask, false, &fr);
    if (j != DD_NOIDX && T.idx[j] == base + i) {                       // this record is the first with its hash: the cell's key is its key
        T.ka[j] = ka;
        T.kb[j] = kb;
    }
}

__global__ __launch_bounds__(256) void k_dd_verdict(DdParams P, DdTable T, DdState *S, DdOver *over, uint8_t *__restrict__ f5_a, DD_ARGS)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    bool drop = false, drop_uniq = false;
    if (i < n) {
        DD_LOAD(i);
        if (ok) {
            const uint32_t me = base + i;
            if (!uniq) {
                drop = S->first_ok != me;
            } else {
                bool fr = false;
                const uint32_t j = dd_cell(T, dd_hash(ka, kb) & P.hash_mask, false, &fr);
                if (T.ka[j] == ka && T.kb[j] == kb) {
                    drop = drop_uniq = T.idx[j] < me;
                } else {
                    // another key owns this hash's cell: settled by full key in k_dd_overflow (the flag is written there)
                    const uint32_t at = atomicAdd(&S->n_over, 1u);
                    if (at < DD_OVER_CAP) over[at] = DdOver{ka, kb, me};
                }
            }
            if (drop) f5_a[i] = (uint8_t)(f5 | F5_NOLOOKUP);
        }
    }
    dd_count(&S->dropped, drop);
    dd_count(&S->dup_unique, drop_uniq);
}

// The overflow list (all windows so far; entries [n_before, n_now) are this window's): an entry is a duplicate iff an entry with
// a smaller record number has the same full key. A handful of entries at most: one workgroup, every thread one new entry.
__global__ __launch_bounds__(256) void k_dd_overflow(DdState *S, const DdOver *over, uint32_t n_before, uint32_t n_now, uint8_t *__restrict__ f5_a, uint32_t base)
{
    for (uint32_t e = n_before + threadIdx.x; e < n_now; e += 256u) {
        const DdOver me = over[e];
        bool dup = false;
        for (uint32_t k = 0; k < n_now; k++)
            if (k != e && over[k].ka == me.ka && over[k].kb == me.kb && over[k].idx < me.idx) dup = true;
        if (dup) {
            const uint32_t i = me.idx - base;
            f5_a[i] = (uint8_t)(f5_a[i] | F5_NOLOOKUP);
            atomicAdd(&S->dropped, 1ull);
            atomicAdd(&S->dup_unique, 1ull);
        }
    }
}

__global__ __launch_bounds__(256) void k_dd_clear(DdTable T, uint32_t cap)
{
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < cap; j += gridDim.x * 256u) {
        T.h[j] = DD_EMPTY;
        T.idx[j] = DD_NOIDX;
    }
}

// every cell of the old table into the new one (hashes are distinct: a claim never meets its own kind)
__global__ __launch_bounds__(256) void k_dd_rehash(DdTable O, uint32_t old_cap, DdTable N)
{
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < old_cap; j += gridDim.x * 256u) {
        const unsigned long long hv = O.h[j];
        if (hv == DD_EMPTY) continue;
        uint32_t k = (uint32_t)((hv * 0x9e3779b97f4a7c15ull) >> 32) & N.mask;
        for (;;) {
            if (atomicCAS(&N.h[k], DD_EMPTY, hv) == DD_EMPTY) break;
            k = (k + 1) & N.mask;
        }
        N.idx[k] = O.idx[j];
        N.ka[k] = O.ka[j];
        N.kb[k] = O.kb[j];
    }
}

struct itx_dedup {
    int device;
    int n_chrom;
    std::vector<int32_t> chrom_size;
    DdParams p;
    DdTable t;
    size_t cap;
    void *d_tid, *d_tid_name;
    DdState *d_state;
    DdOver *d_over;
    hipStream_t st;
    uint64_t base;                         // records seen so far (the next window's first number)
    uint32_t n_over_seen;
};

static int dd_alloc(DdTable *T, size_t cap, hipStream_t st)
{
    memset(T, 0, sizeof *T);
    DD_HIP(hipMalloc((void **)&T->h, cap * 8));
    DD_HIP(hipMalloc((void **)&T->idx, cap * 4));
    DD_HIP(hipMalloc((void **)&T->ka, cap * 8));
    DD_HIP(hipMalloc((void **)&T->kb, cap * 4));
    T->mask = (uint32_t)(cap - 1);
    hipLaunchKernelGGL(k_dd_clear, dim3(4096), dim3(256), 0, st, *T, (uint32_t)cap);
    DD_HIP(hipGetLastError());
    return ITX_OK;
}
static void dd_free(DdTable *T)
{
    (void)hipFree(T->h);
    (void)hipFree(T->idx);
    (void)hipFree(T->ka);
    (void)hipFree(T->kb);
    memset(T, 0, sizeof *T);
}

/* chrom_size[n_chrom] as for itx_table_create; p: mapq_min, extension, isize_max, treat_pe_as_se, discard_half_mapped */
extern "C" int itx_dedup_create(int device, const int64_t *chrom_size, int n_chrom, const itx_params *p, size_t first_cells, itx_dedup **out)
{
    if (!out || !p || n_chrom < 0 || (n_chrom && !chrom_size)) return ITX_E_ARG;
    *out = nullptr;
    DD_HIP(hipSetDevice(device));
    itx_dedup *d = new itx_dedup();
    d->device = device;
    d->n_chrom = n_chrom;
    d->chrom_size.resize((size_t)n_chrom + 1);
    for (int c = 0; c < n_chrom; c++) d->chrom_size[(size_t)c] = (int32_t)chrom_size[c];
    memset(&d->p, 0, sizeof d->p);
    d->p.mapq_min = (uint32_t)p->mapq_min;
    d->p.extension = p->extension;
    d->p.isize_max = p->isize_max;
    d->p.treat = p->treat_pe_as_se;
    d->p.discard = p->discard_half_mapped;
    d->p.hash_mask = ~0ull;
    if (const char *e = getenv("ITX_DEDUP_HASH_BITS"))
        if (atoi(e) >= 8 && atoi(e) < 64) d->p.hash_mask = (1ull << atoi(e)) - 1ull;
    d->d_tid = d->d_tid_name = nullptr;
    d->base = 0;
    d->n_over_seen = 0;
    DD_HIP(hipStreamCreateWithFlags(&d->st, hipStreamNonBlocking));
    size_t cap = 1u << 16;
    while (cap < first_cells && cap < ((size_t)1 << 31)) cap <<= 1;
    d->cap = cap;
    int rc = dd_alloc(&d->t, cap, d->st);
    if (rc != ITX_OK) return rc;
    DD_HIP(hipMalloc((void **)&d->d_state, sizeof(DdState)));
    DD_HIP(hipMalloc((void **)&d->d_over, sizeof(DdOver) * DD_OVER_CAP));
    DdState s0;
    memset(&s0, 0, sizeof s0);
    s0.first_ok = DD_NOIDX;
    DD_HIP(hipMemcpyAsync(d->d_state, &s0, sizeof s0, hipMemcpyHostToDevice, d->st));
    DD_HIP(hipStreamSynchronize(d->st));
    *out = d;
    return ITX_OK;
}

extern "C" void itx_dedup_destroy(itx_dedup *d)
{
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->st) {
        (void)hipStreamSynchronize(d->st);
        (void)hipStreamDestroy(d->st);
    }
    dd_free(&d->t);
    (void)hipFree(d->d_tid);
    (void)hipFree(d->d_tid_name);
    (void)hipFree(d->d_state);
    (void)hipFree(d->d_over);
    delete d;
}

/* the BAM header in use: tid2chrom as for itx_engine_set_tidmap; tid2name[t]: an id of the (renamed) chromosome string that is
 * equal for equal strings across all files of the run (the reference's key holds the string) */
extern "C" int itx_dedup_set_tidmap(itx_dedup *d, const int32_t *tid2chrom, const uint32_t *tid2name, int n_tid)
{
    if (!d || n_tid < 0 || (n_tid && (!tid2chrom || !tid2name))) return ITX_E_ARG;
    DD_HIP(hipSetDevice(d->device));
    std::vector<int2> v((size_t)n_tid + 1);
    std::vector<uint32_t> nm((size_t)n_tid + 1);
    for (int k = 0; k < n_tid; k++) {
        const int32_t c = tid2chrom[k];
        v[(size_t)k] = make_int2(c, (c >= 0 && c < d->n_chrom) ? d->chrom_size[(size_t)c] : 0);
        if (tid2name[k] >= (1u << 31)) {
            itx_set_error("itx_dedup_set_tidmap: name id out of range");
            return ITX_E_ARG;
        }
        nm[(size_t)k] = tid2name[k];
    }
    DD_HIP(hipStreamSynchronize(d->st));
    (void)hipFree(d->d_tid);
    (void)hipFree(d->d_tid_name);
    d->d_tid = d->d_tid_name = nullptr;
    DD_HIP(hipMalloc(&d->d_tid, sizeof(int2) * ((size_t)n_tid + 1)));
    DD_HIP(hipMalloc(&d->d_tid_name, 4 * ((size_t)n_tid + 1)));
    DD_HIP(hipMemcpy(d->d_tid, v.data(), sizeof(int2) * ((size_t)n_tid + 1), hipMemcpyHostToDevice));
    DD_HIP(hipMemcpy(d->d_tid_name, nm.data(), 4 * ((size_t)n_tid + 1), hipMemcpyHostToDevice));
    d->p.tid = (const int2 *)d->d_tid;
    d->p.tid_name = (const uint32_t *)d->d_tid_name;
    d->p.n_tid = n_tid;
    return ITX_OK;
}

/* The next n records of the stream (DEVICE arrays; mpos / isize may be NULL when no record is paired): the dropped ones get
 * ITX_F5_NOLOOKUP in flag5. Synchronous. Call with the records in file order, window after window. */
extern "C" int itx_dedup_run(itx_dedup *d, const int32_t *tid, const int32_t *pos, const int32_t *tmpend, const uint8_t *mapq, uint8_t *flag5, const int32_t *mpos,
                             const int32_t *isize, size_t n)
{
    if (!d || (n && (!tid || !pos || !tmpend || !mapq || !flag5))) return ITX_E_ARG;
    if (!d->p.tid) {
        itx_set_error("itx_dedup_run: no tid map");
        return ITX_E_STATE;
    }
    if (n == 0) return ITX_OK;
    if (d->base + n >= 0xffffffffull || n > 0x7fffffffu) {
        itx_set_error("itx_dedup_run: more than 2^32 records");
        return ITX_E_LIMIT;
    }
    DD_HIP(hipSetDevice(d->device));
    DdState s;
    DD_HIP(hipMemcpyAsync(&s, d->d_state, sizeof s, hipMemcpyDeviceToHost, d->st));
    DD_HIP(hipStreamSynchronize(d->st));
    // room for every record of the window to bring a new key, at a load factor below 0.7
    if (((size_t)s.n_cells + n) * 10 > d->cap * 7) {
        size_t ncap = d->cap;
        while (((size_t)s.n_cells + n) * 10 > ncap * 5) ncap <<= 1;       // grown to below one half, so that growing is rare
        if (ncap > ((size_t)1 << 31)) {
            itx_set_error("itx_dedup_run: the key table would exceed 2^31 cells");
            return ITX_E_LIMIT;
        }
        DdTable nt;
        int rc = dd_alloc(&nt, ncap, d->st);
        if (rc != ITX_OK) return rc;
        hipLaunchKernelGGL(k_dd_rehash, dim3(4096), dim3(256), 0, d->st, d->t, (uint32_t)d->cap, nt);
        DD_HIP(hipGetLastError());
        DD_HIP(hipStreamSynchronize(d->st));
        dd_free(&d->t);
        d->t = nt;
        d->cap = ncap;
    }
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    const uint32_t base = (uint32_t)d->base;
    hipLaunchKernelGGL(k_dd_claim, grid, blk, 0, d->st, d->p, d->t, d->d_state, (const uint8_t *)flag5, tid, pos, tmpend, mapq, mpos, isize, (uint32_t)n, base);
    hipLaunchKernelGGL(k_dd_owner, grid, blk, 0, d->st, d->p, d->t, (const uint8_t *)flag5, tid, pos, tmpend, mapq, mpos, isize, (uint32_t)n, base);
    hipLaunchKernelGGL(k_dd_verdict, grid, blk, 0, d->st, d->p, d->t, d->d_state, d->d_over, flag5, tid, pos, tmpend, mapq, mpos, isize, (uint32_t)n, base);
    DD_HIP(hipGetLastError());
    DD_HIP(hipMemcpyAsync(&s, d->d_state, sizeof s, hipMemcpyDeviceToHost, d->st));
    DD_HIP(hipStreamSynchronize(d->st));
    if (s.n_over > DD_OVER_CAP) {
        itx_set_error("itx_dedup_run: more than %u keys share a 64-bit hash with another key", DD_OVER_CAP);
        return ITX_E_LIMIT;
    }
    if (s.n_over > d->n_over_seen) {
        hipLaunchKernelGGL(k_dd_overflow, dim3(1), blk, 0, d->st, d->d_state, (const DdOver *)d->d_over, d->n_over_seen, s.n_over, flag5, base);
        DD_HIP(hipGetLastError());
        DD_HIP(hipStreamSynchronize(d->st));
        d->n_over_seen = s.n_over;
    }
    d->base += n;
    return ITX_OK;
}

/* so far: records with MAPQ >= Q dropped (what cnt[11] is corrected by), all records dropped, distinct keys */
extern "C" int itx_dedup_counts(itx_dedup *d, uint64_t *dup_unique, uint64_t *dropped, uint64_t *keys)
{
    if (!d) return ITX_E_ARG;
    DD_HIP(hipSetDevice(d->device));
    DdState s;
    DD_HIP(hipMemcpy(&s, d->d_state, sizeof s, hipMemcpyDeviceToHost));
    if (dup_unique) *dup_unique = s.dup_unique;
    if (dropped) *dropped = s.dropped;
    if (keys) *keys = (uint64_t)s.n_cells + s.n_over;
    return ITX_OK;
}
