// itx_partition.hip — ITX_ACCUM_PARTITION: accumulate without scattered global atomics.
//
// Problem: every classified read adds +1 at two consensus slots (range start / range end) of its
// unit. Reads arrive in GENOME order, the slots are in REPEAT-NAME order: a transposition. Done with
// global atomics it costs one memory-side atomic request per key (measured ~9 G/s on MI355X: 10x the
// rest of the path). Here the transposition is done the radix way:
//
//   A  k_stream<EMIT> (itx_stream.hip)  1-2 keys per classified record (slot<<2 | isEnd<<1 | uniq),
//                  written compacted into the workgroup's own region (LDS cursor, no global atomic).
//   C  k_count     per-partition key counts: run lengths per wave (sorted input gives long runs), LDS
//                  histogram per workgroup, one global add per touched partition.
//   S  k_plan      one workgroup: exclusive scan of the counts, scatter cursors, and the work list of
//                  (partition, key range) items for H (a partition with more than ITX_CHUNK keys is split).
//   P  k_scatter   per 4096-key tile: LDS histogram with returning adds (= local ranks), one global
//                  reservation per touched partition, keys written to their partition's range.
//   H  k_hist      per item: LDS window of the partition's W slots (A|B counts, all:16|uniq:16 packed),
//                  ds_add per key, then the window is added into the global A/B arrays — plain
//                  coalesced read-modify-write when the item owns its partition, atomics (few: the
//                  partition is hot, so many keys share a slot) when it is one of several.
//
// Integer sums only: the result is independent of order, identical to the atomic path and the oracle.
#include "itx_partition.h"

#define PB 256              // threads per workgroup
#define ITX_LOGW 13         // slots per partition (W = 8192): LDS window of k_hist = W * 8 bytes = 64 KiB
#define ITX_W (1u << ITX_LOGW)
#define ITX_CHUNK 32768u    // max keys per k_hist item: keeps the packed 16-bit halves from overflowing
#define ITX_TILE 4096u      // keys per k_scatter tile (16 per thread)
#define ITX_MAXP 8192u      // partitions the LDS histograms are sized for

struct ItxPartWork {
    size_t cap;            // records per batch
    uint32_t n_part;       // partitions
    uint32_t max_blocks;   // workgroups of the emit launch (regions are per workgroup)
    uint32_t max_items;
    uint32_t *keys0, *keys1;   // [2*cap]
    uint32_t *blk_cnt;         // [max_blocks] keys emitted by each workgroup
    uint32_t *pcount;          // [n_part]
    uint32_t *pbase;           // [n_part+1]
    uint32_t *cursor;          // [n_part]
    uint4    *items;           // [max_items] (partition, begin, end, exclusive)
    uint32_t *n_items;         // [1]
    void *base;
};

static inline size_t al256(size_t x) { return (x + 255) & ~size_t(255); }

int itx_part_create(const itx_table *t, size_t cap, ItxPartWork **out)
{
    const uint32_t n_part = (t->n_slots + ITX_W - 1) >> ITX_LOGW;
    if (n_part > ITX_MAXP) {
        itx_set_error("partition path: %u consensus slots need %u partitions (> %u); use ITX_ACCUM_ATOMIC", t->n_slots, n_part,
                      ITX_MAXP);
        return ITX_E_LIMIT;
    }
    ItxPartWork *w = new ItxPartWork();
    w->cap = cap;
    w->n_part = n_part ? n_part : 1;
    w->max_blocks = 2048;
    w->max_items = w->n_part + (uint32_t)((2 * cap) / ITX_CHUNK) + 2;
    const size_t kcap = 2 * (cap + ITX_STREAM_TILE) * 4 + 64;
    size_t off = 0;
    const size_t o_k0 = off; off = al256(off + kcap);
    const size_t o_k1 = off; off = al256(off + kcap);
    const size_t o_bc = off; off = al256(off + (size_t)w->max_blocks * 4);
    const size_t o_pc = off; off = al256(off + (size_t)w->n_part * 4);
    const size_t o_pb = off; off = al256(off + ((size_t)w->n_part + 1) * 4);
    const size_t o_cu = off; off = al256(off + (size_t)w->n_part * 4);
    const size_t o_it = off; off = al256(off + (size_t)w->max_items * 16);
    const size_t o_ni = off; off = al256(off + 16);
    char *base = nullptr;
    hipError_t he = hipMalloc((void **)&base, off);
    if (he != hipSuccess) {
        itx_set_error("partition path: hipMalloc(%zu) failed: %s", off, hipGetErrorString(he));
        delete w;
        return ITX_E_NOMEM;
    }
    w->base = base;
    w->keys0 = (uint32_t *)(base + o_k0);
    w->keys1 = (uint32_t *)(base + o_k1);
    w->blk_cnt = (uint32_t *)(base + o_bc);
    w->pcount = (uint32_t *)(base + o_pc);
    w->pbase = (uint32_t *)(base + o_pb);
    w->cursor = (uint32_t *)(base + o_cu);
    w->items = (uint4 *)(base + o_it);
    w->n_items = (uint32_t *)(base + o_ni);
    *out = w;
    return ITX_OK;
}

void itx_part_destroy(ItxPartWork *w)
{
    if (!w) return;
    if (w->base) (void)hipFree(w->base);
    delete w;
}

// ------------------------------------------------------------------------------------------------ C
__global__ __launch_bounds__(PB) void k_count(const uint32_t *__restrict__ keys0, const uint32_t *__restrict__ blk_cnt, size_t span,
                                              uint32_t *__restrict__ pcount, uint32_t n_part)
{
    extern __shared__ uint32_t s_pc[];                        // [n_part]
    for (uint32_t k = threadIdx.x; k < n_part; k += PB) s_pc[k] = 0;
    __syncthreads();
    const uint32_t total = blk_cnt[blockIdx.x];
    const uint32_t *in = keys0 + 2 * (size_t)blockIdx.x * span;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long above = ~(((1ull << lane) - 1ull) << 1 | 1ull);      // lanes strictly above this one
    const uint32_t rounds = (total + PB - 1) / PB;
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t idx = r * PB + threadIdx.x;
        const bool has = idx < total;
        const uint32_t p = has ? in[idx] >> (2 + ITX_LOGW) : 0xffffffffu;
        const uint32_t pp = (uint32_t)__shfl_up((int32_t)p, 1, 64);
        const bool st = has && (lane == 0 || pp != p);          // first lane of a run of equal partitions
        const unsigned long long m_st = __ballot(st), m_has = __ballot(has);
        if (st) {
            const unsigned long long stop = (m_st | ~m_has) & above;
            const uint32_t e = stop ? (uint32_t)__ffsll((long long)stop) - 1u : 64u;
            atomicAdd(&s_pc[p], e - lane);
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n_part; k += PB) {
        const uint32_t v = s_pc[k];
        if (v) atomicAdd(&pcount[k], v);
    }
}

// ------------------------------------------------------------------------------------------------ S
__global__ __launch_bounds__(1024) void k_plan(const uint32_t *__restrict__ pcount, uint32_t n_part, uint32_t *__restrict__ pbase,
                                               uint32_t *__restrict__ cursor, uint4 *__restrict__ items, uint32_t *__restrict__ n_items)
{
    // n_part <= 8192: each of 1024 threads owns up to 8 consecutive partitions
    __shared__ uint32_t s_k[1024], s_i[1024];
    const uint32_t per = (n_part + 1023) / 1024;
    const uint32_t p0 = threadIdx.x * per;
    uint32_t keys = 0, its = 0;
    for (uint32_t p = p0; p < p0 + per && p < n_part; p++) {
        const uint32_t c = pcount[p];
        keys += c;
        its += (c + ITX_CHUNK - 1) / ITX_CHUNK;
    }
    s_k[threadIdx.x] = keys;
    s_i[threadIdx.x] = its;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        uint32_t a = 0, b = 0;
        if (threadIdx.x >= o) {
            a = s_k[threadIdx.x - o];
            b = s_i[threadIdx.x - o];
        }
        __syncthreads();
        s_k[threadIdx.x] += a;
        s_i[threadIdx.x] += b;
        __syncthreads();
    }
    uint32_t kb = s_k[threadIdx.x] - keys, ib = s_i[threadIdx.x] - its;   // exclusive
    for (uint32_t p = p0; p < p0 + per && p < n_part; p++) {
        const uint32_t c = pcount[p];
        pbase[p] = kb;
        cursor[p] = kb;
        const uint32_t ni = (c + ITX_CHUNK - 1) / ITX_CHUNK;
        for (uint32_t j = 0; j < ni; j++) {
            const uint32_t b = kb + j * ITX_CHUNK;
            const uint32_t e = (j + 1 == ni) ? kb + c : b + ITX_CHUNK;
            items[ib + j] = make_uint4(p, b, e, ni == 1 ? 1u : 0u);
        }
        kb += c;
        ib += ni;
    }
    if (threadIdx.x == 1023) {
        pbase[n_part] = s_k[1023];
        *n_items = s_i[1023];
    }
}

// ------------------------------------------------------------------------------------------------ P
__global__ __launch_bounds__(PB) void k_scatter(const uint32_t *__restrict__ keys0, const uint32_t *__restrict__ blk_cnt, size_t span,
                                                uint32_t *__restrict__ cursor, uint32_t *__restrict__ keys1, uint32_t n_part)
{
    extern __shared__ uint32_t s_bin[];                       // [n_part] count, then reused as base
    const uint32_t total = blk_cnt[blockIdx.x];
    const uint32_t *in = keys0 + 2 * (size_t)blockIdx.x * span;
    for (uint32_t k = threadIdx.x; k < n_part; k += PB) s_bin[k] = 0;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < total; t0 += ITX_TILE) {
        uint32_t key[ITX_TILE / PB], rank[ITX_TILE / PB];
#pragma unroll
        for (int j = 0; j < (int)(ITX_TILE / PB); j++) {
            const uint32_t idx = t0 + j * PB + threadIdx.x;
            key[j] = 0xffffffffu;
            rank[j] = 0;
            if (idx < total) {
                key[j] = in[idx];
                rank[j] = atomicAdd(&s_bin[key[j] >> (2 + ITX_LOGW)], 1u);   // ds_add_rtn: rank inside (tile, partition)
            }
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < n_part; k += PB) {
            const uint32_t c = s_bin[k];
            if (c) s_bin[k] = atomicAdd(&cursor[k], c);          // reserve c places in partition k
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < (int)(ITX_TILE / PB); j++)
            if (key[j] != 0xffffffffu) keys1[s_bin[key[j] >> (2 + ITX_LOGW)] + rank[j]] = key[j];
        __syncthreads();
        // clear only what was touched (bins now hold bases; untouched ones are still 0)
#pragma unroll
        for (int j = 0; j < (int)(ITX_TILE / PB); j++)
            if (key[j] != 0xffffffffu) s_bin[key[j] >> (2 + ITX_LOGW)] = 0;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ H
__global__ __launch_bounds__(PB) void k_hist(const uint32_t *__restrict__ keys1, const uint4 *__restrict__ items,
                                             const uint32_t *__restrict__ n_items, uint32_t *__restrict__ u32, ItxAccumLayout L,
                                             uint32_t n_slots)
{
    __shared__ uint32_t s_a[ITX_W], s_b[ITX_W];               // packed all:16 | uniq:16
    const uint32_t nI = *n_items;
    for (uint32_t it = blockIdx.x; it < nI; it += gridDim.x) {
        const uint4 item = items[it];
        const uint32_t slot0 = item.x << ITX_LOGW;
        for (uint32_t k = threadIdx.x; k < ITX_W; k += PB) {
            s_a[k] = 0;
            s_b[k] = 0;
        }
        __syncthreads();
        for (uint32_t k = item.y + threadIdx.x; k < item.z; k += PB) {
            const uint32_t key = keys1[k];
            const uint32_t sl = (key >> 2) - slot0;
            const uint32_t v = 1u | ((key & 1u) << 16);
            if (key & 2u) atomicAdd(&s_b[sl], v); else atomicAdd(&s_a[sl], v);
        }
        __syncthreads();
        uint32_t lim = n_slots - slot0;
        if (lim > ITX_W) lim = ITX_W;
        if (item.w) {
            // this item owns the partition: plain read-modify-write, coalesced
            for (uint32_t k = threadIdx.x; k < lim; k += PB) {
                const uint32_t a = s_a[k], b = s_b[k];
                if (a) {
                    u32[L.a_all + slot0 + k] += a & 0xffffu;
                    if (a >> 16) u32[L.a_uniq + slot0 + k] += a >> 16;
                }
                if (b) {
                    u32[L.b_all + slot0 + k] += b & 0xffffu;
                    if (b >> 16) u32[L.b_uniq + slot0 + k] += b >> 16;
                }
            }
        } else {
            for (uint32_t k = threadIdx.x; k < lim; k += PB) {
                const uint32_t a = s_a[k], b = s_b[k];
                if (a) {
                    atomicAdd(&u32[L.a_all + slot0 + k], a & 0xffffu);
                    if (a >> 16) atomicAdd(&u32[L.a_uniq + slot0 + k], a >> 16);
                }
                if (b) {
                    atomicAdd(&u32[L.b_all + slot0 + k], b & 0xffffu);
                    if (b >> 16) atomicAdd(&u32[L.b_uniq + slot0 + k], b >> 16);
                }
            }
        }
        __syncthreads();
    }
}

int itx_part_run(ItxPartWork *w, const ItxDevTable &T, const ItxRunParams &P, const ItxDevBatch &B, size_t n,
                 int32_t *d_hit_row, uint64_t *u64, uint32_t *u32, const ItxAccumLayout &L, hipStream_t st)
{
    if (n == 0) return ITX_OK;
    if (n > w->cap) {
        itx_set_error("partition path: batch of %zu exceeds capacity %zu", n, w->cap);
        return ITX_E_ARG;
    }
    // every workgroup takes one contiguous span of records (a multiple of the stream tile)
    size_t span = (n + w->max_blocks - 1) / w->max_blocks;
    span = (span + ITX_STREAM_TILE - 1) / ITX_STREAM_TILE * ITX_STREAM_TILE;
    const uint32_t nb = (uint32_t)((n + span - 1) / span);
    ITX_HIP(hipMemsetAsync(w->pcount, 0, (size_t)w->n_part * 4, st));
    int rc = itx_launch_stream(ITX_DO_EMIT, T, P, B, n, span, nb, d_hit_row, u64, u32, L, w->keys0, w->blk_cnt, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_count, dim3(nb), dim3(PB), (size_t)w->n_part * 4, st, w->keys0, w->blk_cnt, span, w->pcount, w->n_part);
    ITX_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_plan, dim3(1), dim3(1024), 0, st, w->pcount, w->n_part, w->pbase, w->cursor, w->items, w->n_items);
    ITX_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_scatter, dim3(nb), dim3(PB), (size_t)w->n_part * 4, st, w->keys0, w->blk_cnt, span, w->cursor, w->keys1,
                       w->n_part);
    ITX_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_hist, dim3(2048), dim3(PB), 0, st, w->keys1, w->items, w->n_items, u32, L, T.n_slots);
    ITX_HIP(hipGetLastError());
    return ITX_OK;
}
