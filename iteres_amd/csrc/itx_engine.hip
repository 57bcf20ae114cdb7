// itx_engine.hip — the C ABI of include/iteres_amd.h: engine life cycle, pinned double buffering,
// submission, partial export (multi-GPU), finish. No CPU fallback anywhere: without a usable HIP device
// every entry point fails.
#include "itx_common.h"
#include "itx_partition.h"

#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

struct ItxSlot {
    itx_staging h;        // pinned host
    itx_staging d;        // device mirrors
    hipStream_t stream;   // copy stream of this slot (H2D in, hit rows D2H out)
    hipEvent_t copied;    // H2D of the slot finished
    hipEvent_t done;      // kernels of the slot finished on the compute stream
    bool busy;
};

struct itx_engine {
    const itx_table *t;
    itx_params p;
    size_t cap;
    ItxAccumLayout L;
    uint64_t *u64;        // cnt[16] | unit_all[n_units] | unit_uniq[n_units]
    uint32_t *u32;        // raw A/B slot arrays + per-locus counts
    ItxTidRec *d_tidrec;
    int n_tid, cap_tid;
    ItxSlot slot[2];
    hipStream_t compute;  // every kernel of the slot path runs here, in submission order: batches never overlap
                          // each other (the partition path's scratch and the exclusive window updates rely on it)
    bool slots_ready;
    // finish-time buffers (device), allocated on first use
    uint64_t *p64;        // own partial
    uint32_t *p32;
    uint64_t *d_counts;
    uint32_t *d_cov, *d_cov_uniq, *d_locus_out;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    double kernel_ms;
    double stage_ms[5];
    uint64_t records, submits, keys;
    ItxPartWork *pw;      // scratch of the partition path
};

static int use_device(const itx_engine *e)
{
    ITX_HIP(hipSetDevice(e->t->device));
    return ITX_OK;
}

static size_t partial_u64(const itx_engine *e) { return 16 + 2 * (size_t)e->t->n_units; }
static size_t partial_u32(const itx_engine *e)
{
    return e->p.mode == ITX_MODE_STAT ? 2 * (size_t)e->t->n_slots : (size_t)e->t->n_rows;
}

extern "C" int itx_engine_create(const itx_table *t, const itx_params *p, size_t batch_capacity, itx_engine **out)
{
    if (!t || !p || !out || batch_capacity == 0) {
        itx_set_error("itx_engine_create: bad argument");
        return ITX_E_ARG;
    }
    if (batch_capacity >= (1ull << 31) - 2 * ITX_STREAM_TILE) {
        itx_set_error("itx_engine_create: batch_capacity %zu too large (< 2^31)", batch_capacity);
        return ITX_E_LIMIT;
    }
    if (p->mode != ITX_MODE_STAT && p->mode != ITX_MODE_FILTER) {
        itx_set_error("itx_engine_create: unknown mode %d", p->mode);
        return ITX_E_ARG;
    }
    if (p->accum != ITX_ACCUM_DEFAULT && p->accum != ITX_ACCUM_ATOMIC && p->accum != ITX_ACCUM_PARTITION) {
        itx_set_error("itx_engine_create: unknown accumulate path %d", p->accum);
        return ITX_E_ARG;
    }
    ITX_HIP(hipSetDevice(t->device));
    itx_engine *e = new (std::nothrow) itx_engine();
    if (!e) return ITX_E_NOMEM;
    e->t = t;
    e->p = *p;
    const bool accum_default = e->p.accum == ITX_ACCUM_DEFAULT;
    if (accum_default) e->p.accum = ITX_ACCUM_PARTITION;
    if (e->p.mode == ITX_MODE_FILTER) e->p.accum = ITX_ACCUM_ATOMIC;   // per-locus counts: one atomic per run of equal rows
    e->cap = batch_capacity;
    e->L = itx_accum_layout(t->n_slots, t->n_rows, t->n_units);
    e->u64 = nullptr;
    e->u32 = nullptr;
    e->d_tidrec = nullptr;
    e->n_tid = e->cap_tid = 0;
    e->slots_ready = false;
    e->compute = nullptr;
    memset(e->slot, 0, sizeof e->slot);
    e->p64 = nullptr;
    e->p32 = nullptr;
    e->d_counts = nullptr;
    e->d_cov = e->d_cov_uniq = e->d_locus_out = nullptr;
    e->kernel_ms = 0;
    e->records = e->submits = e->keys = 0;
    memset(e->stage_ms, 0, sizeof e->stage_ms);
    e->pw = nullptr;
    hipError_t he = hipMalloc((void **)&e->u64, (partial_u64(e) + 2) * sizeof(uint64_t));
    if (he == hipSuccess) he = hipMemset(e->u64, 0, (partial_u64(e) + 2) * sizeof(uint64_t));
    if (he == hipSuccess) he = hipMalloc((void **)&e->u32, (e->L.n_u32 + 4) * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(e->u32, 0, (e->L.n_u32 + 4) * sizeof(uint32_t));
    if (he != hipSuccess) {
        itx_set_error("itx_engine_create: device allocation failed: %s", hipGetErrorString(he));
        itx_engine_destroy(e);
        return ITX_E_NOMEM;
    }
    if (e->p.accum == ITX_ACCUM_PARTITION) {
        int rc = itx_part_create(t, batch_capacity, &e->pw);
        if (rc == ITX_E_LIMIT && accum_default) {
            // a slot space beyond the partition path's widest partitions (> 4096 x 65536 consensus slots): the device atomics
            // path gives the same sums, slower; only an explicit ITX_ACCUM_PARTITION request fails
            e->p.accum = ITX_ACCUM_ATOMIC;
            e->pw = nullptr;
        } else if (rc != ITX_OK) {
            itx_engine_destroy(e);
            return rc;
        }
    }
    *out = e;
    return ITX_OK;
}

static void free_staging(itx_staging *h, itx_staging *d)
{
    void *hp[] = {h->tid, h->pos, h->tmpend, h->mapq, h->flag5, h->mpos, h->isize, h->hit_row};
    void *dp[] = {d->tid, d->pos, d->tmpend, d->mapq, d->flag5, d->mpos, d->isize, d->hit_row};
    for (void *p : hp)
        if (p) (void)hipHostFree(p);
    for (void *p : dp)
        if (p) (void)hipFree(p);
}

extern "C" void itx_engine_destroy(itx_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->t->device);
    (void)hipDeviceSynchronize();
    for (auto &pr : e->ev) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    if (e->slots_ready)
        for (int s = 0; s < 2; s++) {
            free_staging(&e->slot[s].h, &e->slot[s].d);
            if (e->slot[s].stream) (void)hipStreamDestroy(e->slot[s].stream);
            if (e->slot[s].copied) (void)hipEventDestroy(e->slot[s].copied);
            if (e->slot[s].done) (void)hipEventDestroy(e->slot[s].done);
        }
    if (e->compute) (void)hipStreamDestroy(e->compute);
    void *bufs[] = {e->u64, e->u32, e->d_tidrec, e->p64, e->p32, e->d_counts, e->d_cov, e->d_cov_uniq, e->d_locus_out};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (e->pw) itx_part_destroy(e->pw);
    delete e;
}

extern "C" int itx_engine_set_tidmap(itx_engine *e, const int32_t *tid2chrom, int n_tid)
{
    if (!e || n_tid < 0 || (n_tid && !tid2chrom)) {
        itx_set_error("itx_engine_set_tidmap: bad argument");
        return ITX_E_ARG;
    }
    for (int i = 0; i < n_tid; i++)
        if (tid2chrom[i] >= e->t->n_chrom) {
            itx_set_error("itx_engine_set_tidmap: tid %d maps to chromosome %d of %d", i, tid2chrom[i], e->t->n_chrom);
            return ITX_E_ARG;
        }
    int rc = use_device(e);
    if (rc) return rc;
    ITX_HIP(hipDeviceSynchronize());   // earlier batches may still read the previous map
    if (n_tid > e->cap_tid) {
        if (e->d_tidrec) ITX_HIP(hipFree(e->d_tidrec));
        e->d_tidrec = nullptr;
        ITX_HIP(hipMalloc((void **)&e->d_tidrec, sizeof(ItxTidRec) * (size_t)(n_tid + 1)));
        e->cap_tid = n_tid;
    }
    // one 32-byte record per BAM reference id: chromosome, its size, its slice of the table and of the binned index
    std::vector<ItxTidRec> rec((size_t)n_tid);
    const itx_table *t = e->t;
    for (int i = 0; i < n_tid; i++) {
        ItxTidRec &r = rec[i];
        memset(&r, 0, sizeof r);
        const int c = tid2chrom[i];
        r.chrom = c < 0 ? (c == -2 ? -2 : -1) : c;
        if (c >= 0) {
            r.size = t->h_chrom_size[c];
            r.iv_lo = t->h_chrom_off[c];
            r.iv_hi = t->h_chrom_off[c + 1];
            r.bin_base = t->h_bin_off[c];
            if (r.size == 2) r.chrom = -1;             // generic.c:796-797: a size of exactly 2 reads as "not in the size file"
        }
    }
    if (n_tid) ITX_HIP(hipMemcpy(e->d_tidrec, rec.data(), sizeof(ItxTidRec) * (size_t)n_tid, hipMemcpyHostToDevice));
    e->n_tid = n_tid;
    return ITX_OK;
}

static int ensure_compute(itx_engine *e)
{
    if (!e->compute) ITX_HIP(hipStreamCreateWithFlags(&e->compute, hipStreamNonBlocking));
    return ITX_OK;
}

static int ensure_slots(itx_engine *e)
{
    if (e->slots_ready) return ITX_OK;
    const size_t n = e->cap + ITX_STREAM_TILE;
    int rc0 = ensure_compute(e);
    if (rc0) return rc0;
    for (int s = 0; s < 2; s++) {
        ItxSlot &S = e->slot[s];
        ITX_HIP(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
        ITX_HIP(hipEventCreateWithFlags(&S.copied, hipEventDisableTiming));
        ITX_HIP(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
#define BOTH(field, type)                                                                  \
    ITX_HIP(hipHostMalloc((void **)&S.h.field, n * sizeof(type), hipHostMallocDefault));   \
    ITX_HIP(hipMalloc((void **)&S.d.field, n * sizeof(type)));
        BOTH(tid, int32_t)
        BOTH(pos, int32_t)
        BOTH(tmpend, int32_t)
        BOTH(mapq, uint8_t)
        BOTH(flag5, uint8_t)
        BOTH(mpos, int32_t)
        BOTH(isize, int32_t)
        BOTH(hit_row, int32_t)
#undef BOTH
        S.h.capacity = S.d.capacity = e->cap;
        S.busy = false;
    }
    e->slots_ready = true;
    return ITX_OK;
}

extern "C" int itx_engine_staging(itx_engine *e, int slot, itx_staging *out)
{
    if (!e || !out || slot < 0 || slot > 1) {
        itx_set_error("itx_engine_staging: bad argument");
        return ITX_E_ARG;
    }
    int rc = use_device(e);
    if (rc) return rc;
    rc = ensure_slots(e);
    if (rc) return rc;
    *out = e->slot[slot].h;
    return ITX_OK;
}

enum { RUN_ACCUMULATE = 0, RUN_CLASSIFY = 1, RUN_FIND_FIRST = 2 };
static int run_batch(itx_engine *e, const ItxDevBatch &B, size_t n, int32_t *d_hit_row, hipStream_t st, int kind)
{
    if (e->n_tid == 0) {
        itx_set_error("submit before itx_engine_set_tidmap");
        return ITX_E_STATE;
    }
    if (n > e->cap) {
        itx_set_error("submit of %zu records exceeds batch_capacity %zu", n, e->cap);
        return ITX_E_ARG;
    }
    if (n == 0) return ITX_OK;
    ItxRunParams P;
    P.mapq_min = e->p.mapq_min;
    P.min_cov = e->p.min_cov;
    itx_cov_bounds(P.min_cov, &P.cov_hi, &P.cov_lo);
    P.extension = e->p.extension;
    P.isize_max = e->p.isize_max;
    P.treat = e->p.treat_pe_as_se;
    P.discard = e->p.discard_half_mapped;
    P.mode = e->p.mode;
    P.n_tid = e->n_tid;
    P.tidrec = e->d_tidrec;
    hipEvent_t a, b;
    ITX_HIP(hipEventCreate(&a));
    ITX_HIP(hipEventCreate(&b));
    ITX_HIP(hipEventRecord(a, st));
    int rc;
    const bool accumulate = kind == RUN_ACCUMULATE;
    if (accumulate && e->p.accum == ITX_ACCUM_PARTITION) {
        rc = itx_part_run(e->pw, e->t->dev, P, B, n, d_hit_row, e->u64, e->u32, e->L, st);
    } else {
        size_t span;
        unsigned nb;
        itx_stream_plan(itx_stream_blocks(e->t->device, n), n, &span, &nb);
        const int what = kind == RUN_FIND_FIRST ? ITX_DO_FIND_FIRST
                         : !accumulate         ? ITX_DO_CLASSIFY
                                               : (e->p.mode == ITX_MODE_STAT ? ITX_DO_ATOMIC_STAT : ITX_DO_ATOMIC_LOCUS);
        const ItxEmitPlan none = {nullptr, nullptr, 0, 0};
        rc = itx_launch_stream(what, e->t->dev, P, B, n, span, nb, d_hit_row, e->u64, e->u32, e->L, nullptr, nullptr, none, st);
    }
    ITX_HIP(hipEventRecord(b, st));
    e->ev.emplace_back(a, b);
    if (accumulate) {
        e->records += n;
        e->submits++;
    }
    return rc;
}

static int slot_run(itx_engine *e, int slot, size_t n, int has_paired, int want_hits, int kind)
{
    if (!e || slot < 0 || slot > 1) {
        itx_set_error("itx_engine_submit/classify_slot: bad argument");
        return ITX_E_ARG;
    }
    int rc = use_device(e);
    if (rc) return rc;
    rc = ensure_slots(e);
    if (rc) return rc;
    if (n > e->cap) {
        itx_set_error("itx_engine_submit/classify_slot: %zu records exceed the slot capacity %zu", n, e->cap);
        return ITX_E_ARG;
    }
    ItxSlot &S = e->slot[slot];
    if (n == 0) return ITX_OK;
#define H2D(field, type) ITX_HIP(hipMemcpyAsync(S.d.field, S.h.field, n * sizeof(type), hipMemcpyHostToDevice, S.stream));
    H2D(tid, int32_t)
    H2D(pos, int32_t)
    H2D(tmpend, int32_t)
    H2D(mapq, uint8_t)
    H2D(flag5, uint8_t)
    if (has_paired) {
        H2D(mpos, int32_t)
        H2D(isize, int32_t)
    }
#undef H2D
    ItxDevBatch B = {S.d.tid, S.d.pos, S.d.tmpend, S.d.mapq, S.d.flag5, has_paired ? S.d.mpos : nullptr,
                     has_paired ? S.d.isize : nullptr};
    ITX_HIP(hipEventRecord(S.copied, S.stream));
    ITX_HIP(hipStreamWaitEvent(e->compute, S.copied, 0));
    rc = run_batch(e, B, n, want_hits ? S.d.hit_row : nullptr, e->compute, kind);
    if (rc) return rc;
    ITX_HIP(hipEventRecord(S.done, e->compute));
    ITX_HIP(hipStreamWaitEvent(S.stream, S.done, 0));
    if (want_hits)
        ITX_HIP(hipMemcpyAsync(S.h.hit_row, S.d.hit_row, n * sizeof(int32_t), hipMemcpyDeviceToHost, S.stream));
    S.busy = true;
    return ITX_OK;
}

extern "C" int itx_engine_submit_slot(itx_engine *e, int slot, size_t n, int has_paired, int want_hits)
{
    return slot_run(e, slot, n, has_paired, want_hits, RUN_ACCUMULATE);
}

extern "C" int itx_engine_classify_slot(itx_engine *e, int slot, size_t n, int has_paired)
{
    return slot_run(e, slot, n, has_paired, 1, RUN_CLASSIFY);
}

extern "C" int itx_engine_first_hit_slot(itx_engine *e, int slot, size_t n)
{
    return slot_run(e, slot, n, 0, 1, RUN_FIND_FIRST);
}

extern "C" int itx_engine_wait_slot(itx_engine *e, int slot)
{
    if (!e || slot < 0 || slot > 1) {
        itx_set_error("itx_engine_wait_slot: bad argument");
        return ITX_E_ARG;
    }
    if (!e->slots_ready || !e->slot[slot].busy) return ITX_OK;
    int rc = use_device(e);
    if (rc) return rc;
    ITX_HIP(hipStreamSynchronize(e->slot[slot].stream));
    e->slot[slot].busy = false;
    return ITX_OK;
}

static ItxDevBatch to_dev_batch(const itx_batch *b)
{
    ItxDevBatch B = {b->tid, b->pos, b->tmpend, b->mapq, b->flag5, b->mpos, b->isize};
    return B;
}

extern "C" int itx_engine_submit_device(itx_engine *e, const itx_batch *b, size_t n, int32_t *d_hit_row, void *stream)
{
    if (!e || !b || (n && (!b->tid || !b->pos || !b->tmpend || !b->mapq || !b->flag5)) || ((b->mpos == nullptr) != (b->isize == nullptr))) {
        itx_set_error("itx_engine_submit_device: bad argument");
        return ITX_E_ARG;
    }
    int rc = use_device(e);
    if (rc) return rc;
    return run_batch(e, to_dev_batch(b), n, d_hit_row, (hipStream_t)stream, RUN_ACCUMULATE);
}

/* The same on the engine's OWN compute stream — the one the slot submits run on — so that device batches and slot batches of
 * one stream of records stay ordered; itx_engine_wait_own returns when everything submitted there is through (the arrays of
 * `b` may then be reused) without waiting for other work on the device. */
extern "C" int itx_engine_submit_device_own(itx_engine *e, const itx_batch *b, size_t n, int32_t *d_hit_row)
{
    if (!e || !b || (n && (!b->tid || !b->pos || !b->tmpend || !b->mapq || !b->flag5)) || ((b->mpos == nullptr) != (b->isize == nullptr))) {
        itx_set_error("itx_engine_submit_device_own: bad argument");
        return ITX_E_ARG;
    }
    int rc = use_device(e);
    if (rc) return rc;
    rc = ensure_compute(e);                  /* the stream alone: a run that never takes the host route needs no pinned slots */
    if (rc) return rc;
    if (n > e->cap) {
        itx_set_error("itx_engine_submit_device_own: %zu records exceed the batch capacity %zu", n, e->cap);
        return ITX_E_ARG;
    }
    return run_batch(e, to_dev_batch(b), n, d_hit_row, e->compute, RUN_ACCUMULATE);
}

extern "C" int itx_engine_wait_own(itx_engine *e)
{
    if (!e) return ITX_E_ARG;
    int rc = use_device(e);
    if (rc) return rc;
    if (e->compute) ITX_HIP(hipStreamSynchronize(e->compute));
    return ITX_OK;
}

extern "C" int itx_engine_classify_device(itx_engine *e, const itx_batch *b, size_t n, int32_t *d_hit_row, void *stream)
{
    if (!e || !b || !d_hit_row || (n && (!b->tid || !b->pos || !b->tmpend || !b->mapq || !b->flag5))) {
        itx_set_error("itx_engine_classify_device: bad argument");
        return ITX_E_ARG;
    }
    int rc = use_device(e);
    if (rc) return rc;
    return run_batch(e, to_dev_batch(b), n, d_hit_row, (hipStream_t)stream, RUN_CLASSIFY);
}

extern "C" int itx_engine_first_hit_device(itx_engine *e, const itx_batch *b, size_t n, int32_t *d_hit_row, void *stream)
{
    if (!e || !b || !d_hit_row || (n && (!b->tid || !b->pos || !b->tmpend || !b->mapq || !b->flag5))) {
        itx_set_error("itx_engine_first_hit_device: bad argument");
        return ITX_E_ARG;
    }
    int rc = use_device(e);
    if (rc) return rc;
    return run_batch(e, to_dev_batch(b), n, d_hit_row, (hipStream_t)stream, RUN_FIND_FIRST);
}

extern "C" int itx_engine_sync(itx_engine *e)
{
    if (!e) return ITX_E_ARG;
    int rc = use_device(e);
    if (rc) return rc;
    ITX_HIP(hipDeviceSynchronize());
    if (e->slots_ready) e->slot[0].busy = e->slot[1].busy = false;
    return ITX_OK;
}

static void fold_events(itx_engine *e)
{
    for (auto &pr : e->ev) {
        float ms = 0;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess)
            e->kernel_ms += ms;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    e->ev.clear();
    if (e->pw) {
        uint64_t k = 0;
        itx_part_fold_stats(e->pw, e->stage_ms, &k);
        e->keys = k;
    }
}

extern "C" int itx_engine_reset(itx_engine *e)
{
    if (!e) return ITX_E_ARG;
    int rc = itx_engine_sync(e);
    if (rc) return rc;
    ITX_HIP(hipMemset(e->u64, 0, partial_u64(e) * sizeof(uint64_t)));
    ITX_HIP(hipMemset(e->u32, 0, e->L.n_u32 * sizeof(uint32_t)));
    fold_events(e);
    e->kernel_ms = 0;
    e->records = e->submits = e->keys = 0;
    memset(e->stage_ms, 0, sizeof e->stage_ms);
    return ITX_OK;
}

extern "C" int itx_engine_partial_size(const itx_engine *e, uint64_t *n_u64, uint64_t *n_u32)
{
    if (!e) {
        itx_set_error("itx_engine_partial_size: bad argument");
        return ITX_E_ARG;
    }
    if (n_u64) *n_u64 = partial_u64(e);
    if (n_u32) *n_u32 = partial_u32(e);
    return ITX_OK;
}

extern "C" int itx_engine_export_partial(itx_engine *e, void *d_u64, void *d_u32, void *stream)
{
    if (!e || !d_u64 || (!d_u32 && partial_u32(e))) {
        itx_set_error("itx_engine_export_partial: bad argument");
        return ITX_E_ARG;
    }
    int rc = use_device(e);
    if (rc) return rc;
    return itx_launch_export(e->t, e->p.mode, e->u64, e->u32, e->L, (uint64_t *)d_u64, (uint32_t *)d_u32, (hipStream_t)stream);
}

/* the engine's own pair of partial buffers (what itx_engine_finish exports into): for a driver that has no device
 * allocator of its own — export into them, reduce them across ranks, finish from them */
extern "C" int itx_engine_partial_buffers(itx_engine *e, void **d_u64, void **d_u32)
{
    if (!e || !d_u64 || !d_u32) {
        itx_set_error("itx_engine_partial_buffers: bad argument");
        return ITX_E_ARG;
    }
    int rc = use_device(e);
    if (rc) return rc;
    if (!e->p64) {
        ITX_HIP(hipMalloc((void **)&e->p64, sizeof(uint64_t) * (partial_u64(e) + 2)));
        ITX_HIP(hipMalloc((void **)&e->p32, sizeof(uint32_t) * (partial_u32(e) + 4)));
    }
    *d_u64 = e->p64;
    *d_u32 = e->p32;
    return ITX_OK;
}

static int ensure_finish_buffers(itx_engine *e)
{
    const itx_table *t = e->t;
    if (e->d_counts) return ITX_OK;
    ITX_HIP(hipMalloc((void **)&e->d_counts, sizeof(uint64_t) * (2 * ((size_t)t->n_rep + t->n_fam + t->n_cla) + 2)));
    ITX_HIP(hipMalloc((void **)&e->d_cov, sizeof(uint32_t) * (t->cov_len + 2)));
    ITX_HIP(hipMalloc((void **)&e->d_cov_uniq, sizeof(uint32_t) * (t->cov_len + 2)));
    ITX_HIP(hipMalloc((void **)&e->d_locus_out, sizeof(uint32_t) * ((size_t)t->n_rows + 2)));
    return ITX_OK;
}

extern "C" int itx_engine_finish_partial(itx_engine *e, const void *d_u64, const void *d_u32, const itx_result *out)
{
    if (!e || !out || !d_u64 || (!d_u32 && partial_u32(e))) {
        itx_set_error("itx_engine_finish_partial: bad argument");
        return ITX_E_ARG;
    }
    int rc = use_device(e);
    if (rc) return rc;
    ITX_HIP(hipDeviceSynchronize());
    rc = ensure_finish_buffers(e);
    if (rc) return rc;
    const itx_table *t = e->t;
    rc = itx_launch_finish(t, e->p.mode, (const uint64_t *)d_u64, (const uint32_t *)d_u32, e->d_counts, e->d_cov, e->d_cov_uniq,
                           e->d_locus_out, 0);
    if (rc) return rc;
    ITX_HIP(hipDeviceSynchronize());
    if (out->cnt) {
        uint64_t c[16];
        ITX_HIP(hipMemcpy(c, d_u64, sizeof c, hipMemcpyDeviceToHost));
        memcpy(out->cnt, c, 13 * sizeof(uint64_t));
    }
    const size_t R = t->n_rep, F = t->n_fam, Cn = t->n_cla;
    if (out->rep_cnt && R) ITX_HIP(hipMemcpy(out->rep_cnt, e->d_counts, sizeof(uint64_t) * 2 * R, hipMemcpyDeviceToHost));
    if (out->fam_cnt && F) ITX_HIP(hipMemcpy(out->fam_cnt, e->d_counts + 2 * R, sizeof(uint64_t) * 2 * F, hipMemcpyDeviceToHost));
    if (out->cla_cnt && Cn)
        ITX_HIP(hipMemcpy(out->cla_cnt, e->d_counts + 2 * R + 2 * F, sizeof(uint64_t) * 2 * Cn, hipMemcpyDeviceToHost));
    if (e->p.mode == ITX_MODE_STAT) {
        if (out->cov && t->cov_len) ITX_HIP(hipMemcpy(out->cov, e->d_cov, sizeof(uint32_t) * t->cov_len, hipMemcpyDeviceToHost));
        if (out->cov_uniq && t->cov_len)
            ITX_HIP(hipMemcpy(out->cov_uniq, e->d_cov_uniq, sizeof(uint32_t) * t->cov_len, hipMemcpyDeviceToHost));
        if (out->locus_cnt && t->n_rows) memset(out->locus_cnt, 0, sizeof(uint32_t) * (size_t)t->n_rows);
    } else {
        if (out->locus_cnt && t->n_rows)
            ITX_HIP(hipMemcpy(out->locus_cnt, e->d_locus_out, sizeof(uint32_t) * (size_t)t->n_rows, hipMemcpyDeviceToHost));
        if (out->cov && t->cov_len) memset(out->cov, 0, sizeof(uint32_t) * t->cov_len);
        if (out->cov_uniq && t->cov_len) memset(out->cov_uniq, 0, sizeof(uint32_t) * t->cov_len);
    }
    return ITX_OK;
}

extern "C" int itx_engine_finish(itx_engine *e, const itx_result *out)
{
    if (!e || !out) {
        itx_set_error("itx_engine_finish: bad argument");
        return ITX_E_ARG;
    }
    int rc = itx_engine_sync(e);
    if (rc) return rc;
    if (!e->p64) {
        ITX_HIP(hipMalloc((void **)&e->p64, sizeof(uint64_t) * (partial_u64(e) + 2)));
        ITX_HIP(hipMalloc((void **)&e->p32, sizeof(uint32_t) * (partial_u32(e) + 4)));
    }
    rc = itx_launch_export(e->t, e->p.mode, e->u64, e->u32, e->L, e->p64, e->p32, 0);
    if (rc) return rc;
    return itx_engine_finish_partial(e, e->p64, e->p32, out);
}

extern "C" int itx_engine_get_stats(itx_engine *e, itx_stats *out)
{
    if (!e || !out) return ITX_E_ARG;
    int rc = use_device(e);
    if (rc) return rc;
    fold_events(e);
    memset(out, 0, sizeof *out);
    out->kernel_ms = e->kernel_ms;
    out->records = e->records;
    out->submits = e->submits;
    out->keys = e->keys;
    memcpy(out->stage_ms, e->stage_ms, sizeof e->stage_ms);
    uint64_t c[16];
    ITX_HIP(hipMemcpy(c, e->u64, sizeof c, hipMemcpyDeviceToHost));
    out->hits = c[9];
    return ITX_OK;
}
