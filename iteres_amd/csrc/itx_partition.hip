// itx_partition.hip — ITX_ACCUM_PARTITION: accumulate without scattered global atomics.
//
// Problem: every classified read adds +1 at two consensus slots (range start / range end) of its
// unit. Reads arrive in GENOME order, the slots are in REPEAT-NAME order: a transposition. Done with
// global atomics it costs one memory-side atomic request per key (measured ~9 G/s on MI355X: 10x the
// rest of the path). Here the transposition is done the radix way:
//
//   A  k_stream<EMIT> (itx_stream.hip)  one 8-byte key per classified record — (type | uniq<<2 | len<<3, slot): a
//                  start mark at slot and an end mark at slot+len — or two when the marks fall into different
//                  partitions, written compacted in record order into the workgroup's own region (LDS cursor, no
//                  global atomic).
//   C  (inside A)  per workgroup region: per-partition key counts (run lengths per wave — sorted input
//                  gives long runs — into an LDS histogram), then ONE reservation per touched partition on
//                  one of 8 sub-cursors (workgroup id mod 8: the dispatcher deals workgroups round-robin over
//                  the 8 XCDs, so a sub-cursor is mostly hit from one XCD and no address sees more than
//                  n_blocks/8 adds); the offsets it got are stored as the region's row of an offset matrix.
//   S  (inside P)  exclusive scan of the 8*P sub-totals -> bases, recomputed by every workgroup of k_scatter (a few tens
//                  of kilobytes out of L2; cheaper than a launch of its own), and the work list of (partition, key
//                  range) items for H (a partition with more than ITX_CHUNK keys is split), written by workgroup 0.
//   P  k_scatter   per region: cursor[p] = base[p][sub] + offset row in LDS, then every run of equal partitions
//                  takes its places with one returning LDS add and writes its keys, now 4 bytes each (slot within
//                  the partition << 16 | low half of the 8-byte key) — no global atomics.
//   H  k_hist      per item: LDS window of the partition's W slots (start | end marks, all:16|uniq:16 packed),
//                  ds_add per key; reads per unit = sums of the start marks over each unit's slots (few u64 atomics);
//                  then D = starts - ends is added into the two global arrays — plain coalesced read-modify-write
//                  when the item owns its partition, atomics (few: the partition is hot, so many keys share a
//                  slot) when it is one of several.
//
// Integer sums only: the result is independent of order, identical to the atomic path and the oracle.
#include "itx_partition.h"
#include "itx_device.h"

#include <vector>

#define PB 256              // threads per workgroup (count / scatter)
#define HB 1024             // threads per workgroup (hist): 2 workgroups per CU keep 32 waves in flight
#define ITX_W (1u << ITX_LOGW)
#define ITX_CHUNK 65535u    // max keys per k_hist item: keeps the packed 16-bit halves from overflowing
#define ITX_MAXP 4096u      // partitions the LDS histograms / the plan kernel are sized for
#define ITX_SUB ITX_PART_SUB // sub-cursors per partition

struct ItxPartWork {
    size_t cap;            // records per batch
    uint32_t n_part;       // partitions
    uint32_t log_w;        // log2(slots per partition): ITX_LOGW .. 16 (the smallest that keeps n_part <= ITX_MAXP)
    uint32_t max_blocks;   // workgroups of the emit launch (regions are per workgroup)
    int device;
    uint32_t max_items;
    uint2 *keys0;              // [2*cap] 8-byte keys as emitted, per workgroup region
    uint32_t *keys1;           // [2*cap] 4-byte keys, partitioned
    uint32_t *blk_cnt;         // [max_blocks][4] keys emitted by each wave of each workgroup (into its quarter of the region)
    uint32_t *subcur;          // [n_part*8] keys reserved per (partition, sub-cursor); zero between batches (k_hist clears it)
    uint32_t *offm;            // [max_blocks][n_part] offset of each region inside (partition, sub)
    uint4    *items;           // [max_items] (partition, begin, end, exclusive)
    uint32_t *n_items;         // [1]
    void *base;
    std::vector<hipEvent_t> ev;    // 4 events per batch: before stream, after stream, after scatter, after hist
    uint64_t keys_last;
};

static inline size_t al256(size_t x) { return (x + 255) & ~size_t(255); }

int itx_part_create(const itx_table *t, size_t cap, ItxPartWork **out)
{
    // A partition is one or more LDS windows of k_hist (2^ITX_LOGW slots each): as many as keep the partitions within what
    // the per-region LDS tables of k_stream / k_scatter hold. Up to 2^16 slots per partition the 4-byte keys of k_scatter
    // still carry slot-in-partition (16 bits) + length (13) + type/uniq (3).
    uint32_t log_w = ITX_LOGW;
    if (const char *s = getenv("ITX_PART_LOGW")) {                         // tests: wide partitions on a small slot space
        const long v = atol(s);
        if (v >= (long)ITX_LOGW && v <= 16) log_w = (uint32_t)v;
    }
    while (log_w < 16 && (((uint64_t)t->n_slots + (1ull << log_w) - 1) >> log_w) > ITX_MAXP) log_w++;
    const uint32_t n_part = (uint32_t)(((uint64_t)t->n_slots + (1ull << log_w) - 1) >> log_w);
    if (n_part > ITX_MAXP) {
        itx_set_error("partition path: %u consensus slots need %u partitions of 65536 (> %u); use ITX_ACCUM_ATOMIC", t->n_slots, n_part,
                      ITX_MAXP);
        return ITX_E_LIMIT;
    }
    ItxPartWork *w = new ItxPartWork();
    w->cap = cap;
    w->n_part = n_part ? n_part : 1;
    w->log_w = log_w;
    w->device = t->device;
    w->max_blocks = itx_stream_blocks(t->device, cap);
    w->max_items = w->n_part + (uint32_t)((2 * cap) / ITX_CHUNK) + 2;
    const size_t kcap = 2 * (cap + ITX_STREAM_TILE) * 4 + 64;
    // keys as emitted: every workgroup owns the 2 * span slots of its span of records, its four waves a quarter each, so
    // the slots run to 2 * n_blocks * span — up to one span beyond the records
    // (sized for the longest regions a run may use: a batch below the capacity takes fewer, never longer ones than one round's)
    size_t span_cap;
    unsigned nb_cap;
    itx_stream_plan(itx_stream_blocks(t->device, 1), cap, &span_cap, &nb_cap);
    const size_t k0cap = 2 * (cap + span_cap + ITX_STREAM_TILE) * 8 + 64;
    size_t off = 0;
    const size_t o_k0 = off; off = al256(off + k0cap);
    const size_t o_k1 = off; off = al256(off + kcap);
    const size_t o_bc = off; off = al256(off + (size_t)w->max_blocks * 4 * 4);
    const size_t o_sc = off; off = al256(off + ((size_t)w->n_part * ITX_SUB + 1) * 4);
    const size_t o_om = off; off = al256(off + (size_t)w->max_blocks * w->n_part * 4);
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
    if (hipMemset(base + o_sc, 0, ((size_t)w->n_part * ITX_SUB + 1) * 4) != hipSuccess) {      // k_hist keeps it zero between batches
        itx_set_error("partition path: hipMemset failed");
        (void)hipFree(base);
        delete w;
        return ITX_E_NO_DEVICE;
    }
    w->keys0 = (uint2 *)(base + o_k0);
    w->keys1 = (uint32_t *)(base + o_k1);
    w->blk_cnt = (uint32_t *)(base + o_bc);
    w->subcur = (uint32_t *)(base + o_sc);
    w->offm = (uint32_t *)(base + o_om);
    w->items = (uint4 *)(base + o_it);
    w->n_items = (uint32_t *)(base + o_ni);
    *out = w;
    return ITX_OK;
}

void itx_part_destroy(ItxPartWork *w)
{
    if (!w) return;
    if (w->base) (void)hipFree(w->base);
    for (hipEvent_t e : w->ev) (void)hipEventDestroy(e);
    delete w;
}

// ------------------------------------------------------------------------------------------------ P
// Block-wide exclusive prefix sums of two values per thread (PB threads). Returns the exclusive sums; *tot_* = block totals.
__device__ __forceinline__ void block_excl_scan2(uint32_t a, uint32_t b, uint32_t *ea, uint32_t *eb, uint32_t *tot_a, uint32_t *tot_b)
{
    __shared__ uint32_t s_wa[PB / 64], s_wb[PB / 64];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t ia = a, ib = b;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t ta = (uint32_t)__shfl_up((int32_t)ia, o, 64), tb = (uint32_t)__shfl_up((int32_t)ib, o, 64);
        if (lane >= (uint32_t)o) {
            ia += ta;
            ib += tb;
        }
    }
    if (lane == 63) {
        s_wa[w] = ia;
        s_wb[w] = ib;
    }
    __syncthreads();
    uint32_t oa = 0, ob = 0, ta = 0, tb = 0;
#pragma unroll
    for (uint32_t k = 0; k < PB / 64; k++) {
        if (k < w) {
            oa += s_wa[k];
            ob += s_wb[k];
        }
        ta += s_wa[k];
        tb += s_wb[k];
    }
    *ea = oa + ia - a;
    *eb = ob + ib - b;
    *tot_a = ta;
    *tot_b = tb;
    __syncthreads();
}

__global__ __launch_bounds__(PB) void k_scatter(const uint2 *__restrict__ keys0, const uint32_t *__restrict__ blk_cnt, size_t span,
                                                const uint32_t *__restrict__ subcur, const uint32_t *__restrict__ offm,
                                                uint32_t *__restrict__ keys1, uint32_t n_part, uint32_t log_w,
                                                uint4 *__restrict__ items, uint32_t *__restrict__ n_items)
{
    extern __shared__ uint32_t s_cur[];                       // [n_part] next free place of this region in each partition
    const uint32_t sub = blockIdx.x & (ITX_SUB - 1);
    const uint32_t *row = offm + (size_t)blockIdx.x * n_part;
    // The plan, recomputed by every workgroup (8*P sub-totals, a few tens of kilobytes out of L2): partition p starts at
    // the sum of all smaller partitions' keys, its sub-cursor `sub` behind the sub-totals below it, this region at the
    // offset it reserved there. Workgroup 0 also writes the work list of k_hist: (partition, key range) items, a
    // partition with more than ITX_CHUNK keys split.
    {
        const uint32_t per = (n_part + PB - 1) / PB;                       // <= ITX_MAXP / PB = 16 partitions per thread
        const uint32_t p0 = threadIdx.x * per, p1 = p0 + per < n_part ? p0 + per : n_part;
        uint32_t keys = 0, its = 0, cnt[ITX_MAXP / PB];
#pragma unroll
        for (uint32_t i = 0; i < ITX_MAXP / PB; i++) {
            const uint32_t p = p0 + i;
            cnt[i] = 0;
            if (i < per && p < p1) {
                const uint4 lo = *reinterpret_cast<const uint4 *>(&subcur[p * ITX_SUB]), hi = *reinterpret_cast<const uint4 *>(&subcur[p * ITX_SUB + 4]);
                const uint32_t v[ITX_SUB] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                uint32_t below = 0, c = 0;
#pragma unroll
                for (uint32_t x = 0; x < ITX_SUB; x++) {
                    below += x < sub ? v[x] : 0u;
                    c += v[x];
                }
                cnt[i] = c;
                s_cur[p] = below + row[p];
                keys += c;
                its += (c + ITX_CHUNK - 1) / ITX_CHUNK;
            }
        }
        uint32_t kb, ib, tk, ti;
        block_excl_scan2(keys, its, &kb, &ib, &tk, &ti);
#pragma unroll
        for (uint32_t i = 0; i < ITX_MAXP / PB; i++) {
            const uint32_t p = p0 + i;
            if (i < per && p < p1) {
                const uint32_t c = cnt[i];
                s_cur[p] += kb;
                if (blockIdx.x == 0) {
                    const uint32_t ni = (c + ITX_CHUNK - 1) / ITX_CHUNK;
                    for (uint32_t j = 0; j < ni; j++) {
                        const uint32_t b = kb + j * ITX_CHUNK;
                        const uint32_t e = (j + 1 == ni) ? kb + c : b + ITX_CHUNK;
                        items[ib + j] = make_uint4(p, b, e, ni == 1 ? 1u : 0u);
                    }
                    ib += ni;
                }
                kb += c;
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) *n_items = ti;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wmask = (1u << log_w) - 1u;
    // 8 rounds of 256 keys per iteration: the loads of all eight are in flight before the first is used
    // (one round after the other leaves the kernel waiting on one global load per 256 keys per workgroup)
    constexpr int U = 8;
    for (uint32_t q = 0; q < 4; q++) {                         // the four waves of the emitting workgroup filled a quarter each
        const uint32_t total = blk_cnt[4 * blockIdx.x + q];
        const uint2 *in = keys0 + 2 * (size_t)blockIdx.x * span + (size_t)q * (span / 2);
        // software-pipelined: the loads of the next 2048 keys are in flight while the current ones find their places
        uint2 nxt[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t idx = u * PB + threadIdx.x;
            nxt[u] = idx < total ? in[idx] : make_uint2(0, 0xffffffffu);
        }
        for (uint32_t r0 = 0; r0 < total; r0 += U * PB) {
            uint2 key[U];
#pragma unroll
            for (int u = 0; u < U; u++) key[u] = nxt[u];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t idx = r0 + U * PB + u * PB + threadIdx.x;
                nxt[u] = idx < total ? in[idx] : make_uint2(0, 0xffffffffu);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool has = key[u].y != 0xffffffffu;
                const uint32_t p = has ? key[u].y >> log_w : 0xffffffffu;
                uint32_t len, leader;
                const bool st = wave_run(p, has, lane, &len, &leader);
                uint32_t base = 0;
                if (st) base = atomicAdd(&s_cur[p], len);             // one returning LDS add per run
                base = (uint32_t)__shfl((int32_t)base, (int)leader, 64);
                if (has) keys1[base + (lane - leader)] = ((key[u].y & wmask) << 16) | (key[u].x & 0xffffu);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ H
// WIDE: a partition spans several LDS windows (log_w > ITX_LOGW, slot spaces beyond ITX_MAXP x 2^ITX_LOGW): the item's keys
// are read once per window and each pass keeps the marks that fall into its window.
template <bool WIDE>
__global__ __launch_bounds__(HB) void k_hist(const uint32_t *__restrict__ keys1, const uint4 *__restrict__ items,
                                             const uint32_t *__restrict__ n_items, uint32_t *__restrict__ u32, uint64_t *__restrict__ u64,
                                             ItxAccumLayout L, uint32_t n_slots, const uint32_t *__restrict__ unit_slot,
                                             const uint32_t *__restrict__ part_unit, uint32_t *__restrict__ subcur, uint32_t n_sub,
                                             uint32_t log_w)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_a[ITX_W], s_b[ITX_W];   // packed all:16 | uniq:16
    // the sub-totals have done their job (k_scatter is through): zero them for the next batch's reservations
    for (uint32_t i = blockIdx.x * HB + threadIdx.x; i < n_sub; i += gridDim.x * HB) subcur[i] = 0;
    const uint32_t nI = *n_items;
    for (uint32_t it = blockIdx.x; it < nI; it += gridDim.x) {
        const uint4 item = items[it];
        const uint32_t n_win = WIDE ? 1u << (log_w - ITX_LOGW) : 1u;
        for (uint32_t h = 0; h < n_win; h++) {
        const uint32_t off = h << ITX_LOGW;                          // the window's first slot inside the partition
        const uint32_t slot0 = WIDE ? (item.x << log_w) + off : item.x << ITX_LOGW;
        if (WIDE && slot0 >= n_slots) break;                         // the last partition may end early
        for (uint32_t k = threadIdx.x; k < ITX_W; k += HB) {
            s_a[k] = 0;
            s_b[k] = 0;
        }
        __syncthreads();
        for (uint32_t k0 = item.y; k0 < item.z; k0 += 8 * HB) {          // 8 loads in flight per thread
            uint32_t key[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t k = k0 + u * HB + threadIdx.x;
                key[u] = k < item.z ? keys1[k] : 0xffffffffu;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (key[u] != 0xffffffffu) {
                    const uint32_t sl = key[u] >> 16, type = key[u] & 3u, len = (key[u] >> 3) & (ITX_W - 1);
                    const uint32_t v = 1u | ((key[u] & 4u) << 14);
                    if (WIDE) {
                        const uint32_t a = sl - off, b = sl + len - off;         // below the window: wraps to a huge number
                        if (type != 2u && a < ITX_W) atomicAdd(&s_a[a], v);
                        if (type != 1u && b < ITX_W) atomicAdd(&s_b[b], v);
                    } else {
                        if (type != 2u) atomicAdd(&s_a[sl], v);                  // start mark
                        if (type != 1u) atomicAdd(&s_b[sl + len], v);            // end mark (len == 0 for type 2)
                    }
                }
            }
        }
        __syncthreads();
        uint32_t lim = n_slots - slot0;
        if (lim > ITX_W) lim = ITX_W;
        // reads per unit: every classified read left exactly one start mark inside its unit's slots (in the unit's extra
        // slot when it adds no coverage); waves take the units that reach into this window in turn
        {
            const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
            for (uint32_t u = part_unit[WIDE ? (item.x << (log_w - ITX_LOGW)) + h : item.x] + wave; u < (uint32_t)L.n_units; u += HB / 64) {
                const uint32_t us = unit_slot[u], ue = unit_slot[u + 1];
                if (us >= slot0 + lim) break;
                const uint32_t lo = us > slot0 ? us - slot0 : 0u, hi = ue < slot0 + lim ? ue - slot0 : lim;
                uint32_t acc = 0;                                 // packed all:16 | uniq:16; an item holds <= 65535 keys
                for (uint32_t k = lo + lane; k < hi; k += 64) acc += s_a[k];
                for (int o = 32; o > 0; o >>= 1) acc += (uint32_t)__shfl_down((int32_t)acc, o, 64);
                if (lane == 0 && acc) {
                    atomicAdd((unsigned long long *)&u64[16 + u], (unsigned long long)(acc & 0xffffu));
                    if (acc >> 16) atomicAdd((unsigned long long *)&u64[16 + L.n_units + u], (unsigned long long)(acc >> 16));
                }
            }
        }
        if (item.w) {
            // this item owns the partition: plain read-modify-write of D = starts - ends, 4 consecutive slots per thread
            // (16-byte accesses; the arrays start 256-byte aligned and slot0 is a multiple of the window)
            for (uint32_t k = 4 * threadIdx.x; k < ITX_W; k += 4 * HB) {
                const uint4 a = *reinterpret_cast<const uint4 *>(&s_a[k]);
                const uint4 b = *reinterpret_cast<const uint4 *>(&s_b[k]);
                // per slot: low half all reads, high half unique reads; the differences wrap mod 2^32 like the counters
                const uint4 da = make_uint4((a.x & 0xffffu) - (b.x & 0xffffu), (a.y & 0xffffu) - (b.y & 0xffffu), (a.z & 0xffffu) - (b.z & 0xffffu),
                                            (a.w & 0xffffu) - (b.w & 0xffffu));
                const uint4 du = make_uint4((a.x >> 16) - (b.x >> 16), (a.y >> 16) - (b.y >> 16), (a.z >> 16) - (b.z >> 16), (a.w >> 16) - (b.w >> 16));
                const bool anya = (da.x | da.y | da.z | da.w) != 0, anyu = (du.x | du.y | du.z | du.w) != 0;
                if (!(anya || anyu)) continue;
                const size_t g = (size_t)slot0 + k;
                if (k + 4 <= lim) {
                    uint4 xa = make_uint4(0, 0, 0, 0), xu = xa;
                    if (anya) xa = *reinterpret_cast<const uint4 *>(&u32[L.d_all + g]);
                    if (anyu) xu = *reinterpret_cast<const uint4 *>(&u32[L.d_uniq + g]);
                    if (anya) {
                        xa.x += da.x; xa.y += da.y; xa.z += da.z; xa.w += da.w;
                        *reinterpret_cast<uint4 *>(&u32[L.d_all + g]) = xa;
                    }
                    if (anyu) {
                        xu.x += du.x; xu.y += du.y; xu.z += du.z; xu.w += du.w;
                        *reinterpret_cast<uint4 *>(&u32[L.d_uniq + g]) = xu;
                    }
                } else {                                         // the last, partial group of the slot space
                    const uint32_t av[4] = {da.x, da.y, da.z, da.w}, uv[4] = {du.x, du.y, du.z, du.w};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if (k + i < lim) {
                            if (av[i]) u32[L.d_all + g + i] += av[i];
                            if (uv[i]) u32[L.d_uniq + g + i] += uv[i];
                        }
                    }
                }
            }
        } else {
            // one of several items of a hot partition: many keys per slot, so few distinct slots per key — atomics,
            // lane-contiguous so that a wave's adds fall into as few 64-byte requests as possible
            for (uint32_t k = threadIdx.x; k < lim; k += HB) {
                const uint32_t a = s_a[k], b = s_b[k];
                const uint32_t da = (a & 0xffffu) - (b & 0xffffu), du = (a >> 16) - (b >> 16);
                if (da) atomicAdd(&u32[L.d_all + slot0 + k], da);
                if (du) atomicAdd(&u32[L.d_uniq + slot0 + k], du);
            }
        }
        __syncthreads();
        }
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
    unsigned blocks = itx_stream_blocks(w->device, n);
    if (blocks > w->max_blocks) blocks = w->max_blocks;
    size_t span;
    unsigned nb;
    itx_stream_plan(blocks, n, &span, &nb);
    hipEvent_t ev[4];
    for (int k = 0; k < 4; k++) ITX_HIP(hipEventCreate(&ev[k]));
    ITX_HIP(hipEventRecord(ev[0], st));
    ItxEmitPlan E = {w->subcur, w->offm, w->n_part, w->log_w};
    int rc = itx_launch_stream(ITX_DO_EMIT, T, P, B, n, span, nb, d_hit_row, u64, u32, L, w->keys0, w->blk_cnt, E, st);
    if (rc) return rc;
    ITX_HIP(hipEventRecord(ev[1], st));
    hipLaunchKernelGGL(k_scatter, dim3(nb), dim3(PB), (size_t)w->n_part * 4, st, w->keys0, w->blk_cnt, span, w->subcur, w->offm, w->keys1,
                       w->n_part, w->log_w, w->items, w->n_items);
    ITX_HIP(hipGetLastError());
    ITX_HIP(hipEventRecord(ev[2], st));
    static const unsigned hist_blocks = [] {
        const char *s = getenv("ITX_HIST_BLOCKS");
        const long v = s ? atol(s) : 0;
        return v >= 64 && v <= 65536 ? (unsigned)v : 4096u;   // items are dealt round robin: enough workgroups that the hardware balances them (500 M records: 1024 -> 0.96 ms, 4096 -> 0.88)
    }();
    if (w->log_w > ITX_LOGW)
        hipLaunchKernelGGL(k_hist<true>, dim3(hist_blocks), dim3(HB), 0, st, w->keys1, w->items, w->n_items, u32, u64, L, T.n_slots, T.unit_slot,
                           T.part_unit, w->subcur, w->n_part * ITX_SUB, w->log_w);
    else
        hipLaunchKernelGGL(k_hist<false>, dim3(hist_blocks), dim3(HB), 0, st, w->keys1, w->items, w->n_items, u32, u64, L, T.n_slots, T.unit_slot,
                           T.part_unit, w->subcur, w->n_part * ITX_SUB, w->log_w);
    ITX_HIP(hipGetLastError());
    ITX_HIP(hipEventRecord(ev[3], st));
    for (int k = 0; k < 4; k++) w->ev.push_back(ev[k]);
    return ITX_OK;
}

void itx_part_fold_stats(ItxPartWork *w, double *ms, uint64_t *keys_last)
{
    static const int stage_of[3] = {0, 2, 3};                 // stream, (the plan is part of the scatter now), scatter, hist
    for (size_t i = 0; i + 4 <= w->ev.size(); i += 4) {
        if (hipEventSynchronize(w->ev[i + 3]) == hipSuccess)
            for (int k = 0; k < 3; k++) {
                float t = 0;
                if (hipEventElapsedTime(&t, w->ev[i + k], w->ev[i + k + 1]) == hipSuccess) ms[stage_of[k]] += t;
            }
        for (int k = 0; k < 4; k++) (void)hipEventDestroy(w->ev[i + k]);
    }
    w->ev.clear();
    // keys of the most recent batch = end of the last sub-cursor base + ... : read the plan's item list tail instead
    uint32_t ni = 0;
    if (hipMemcpy(&ni, w->n_items, 4, hipMemcpyDeviceToHost) == hipSuccess && ni > 0 && ni <= w->max_items) {
        uint4 last;
        if (hipMemcpy(&last, w->items + (ni - 1), sizeof last, hipMemcpyDeviceToHost) == hipSuccess) w->keys_last = last.z;
    }
    if (keys_last) *keys_last = w->keys_last;
}
