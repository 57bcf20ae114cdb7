// itx_engine.hip — the C ABI of include/iteres_amd.h: engine life cycle, pinned double buffering,
// submission, finish. No CPU fallback anywhere: without a usable HIP device every entry point fails.
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
    uint64_t *u64;
    uint32_t *u32;
    bool own_u64, own_u32;
    int32_t *d_tidmap;
    int n_tid, cap_tid;
    ItxSlot slot[2];
    hipStream_t compute;  // every kernel of the slot path runs here, in submission order: batches never overlap
                          // each other (the partition path's scratch and the exclusive window updates rely on it)
    bool slots_ready;
    // finalize outputs (device), allocated on first finish
    uint64_t *d_rep_out;
    uint32_t *d_cov, *d_cov_uniq, *d_locus_out;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    double kernel_ms;
    uint64_t records;
    ItxPartWork *pw;      // scratch of the partition path
};

static int use_device(const itx_engine *e)
{
    ITX_HIP(hipSetDevice(e->t->device));
    return ITX_OK;
}

extern "C" int itx_engine_create(const itx_table *t, const itx_params *p, size_t batch_capacity, void *u64_accum,
                                 void *u32_accum, itx_engine **out)
{
    if (!t || !p || !out || batch_capacity == 0) {
        itx_set_error("itx_engine_create: bad argument");
        return ITX_E_ARG;
    }
    if (batch_capacity >= (1ull << 31)) {
        itx_set_error("itx_engine_create: batch_capacity %zu >= 2^31", batch_capacity);
        return ITX_E_LIMIT;
    }
    if (p->mode != ITX_MODE_STAT && p->mode != ITX_MODE_FILTER) {
        itx_set_error("itx_engine_create: unknown mode %d", p->mode);
        return ITX_E_ARG;
    }
    ITX_HIP(hipSetDevice(t->device));
    itx_engine *e = new (std::nothrow) itx_engine();
    if (!e) return ITX_E_NOMEM;
    e->t = t;
    e->p = *p;
    if (e->p.accum == ITX_ACCUM_DEFAULT) e->p.accum = ITX_ACCUM_PARTITION;
    if (e->p.mode == ITX_MODE_FILTER) e->p.accum = ITX_ACCUM_ATOMIC;   // per-locus counts: one aggregated atomic per run
    e->cap = batch_capacity;
    e->L = itx_accum_layout(t->n_rep, t->n_fam, t->n_cla, t->n_slots, t->n_rows);
    e->u64 = (uint64_t *)u64_accum;
    e->u32 = (uint32_t *)u32_accum;
    e->own_u64 = e->own_u32 = false;
    e->d_tidmap = nullptr;
    e->n_tid = e->cap_tid = 0;
    e->slots_ready = false;
    e->compute = nullptr;
    memset(e->slot, 0, sizeof e->slot);
    e->d_rep_out = nullptr;
    e->d_cov = e->d_cov_uniq = e->d_locus_out = nullptr;
    e->kernel_ms = 0;
    e->records = 0;
    e->pw = nullptr;
    hipError_t he;
    if (!e->u64) {
        he = hipMalloc((void **)&e->u64, e->L.n_u64 * sizeof(uint64_t));
        if (he != hipSuccess) goto nomem;
        e->own_u64 = true;
        he = hipMemset(e->u64, 0, e->L.n_u64 * sizeof(uint64_t));
        if (he != hipSuccess) goto nomem;
    }
    if (!e->u32) {
        he = hipMalloc((void **)&e->u32, e->L.n_u32 * sizeof(uint32_t));
        if (he != hipSuccess) goto nomem;
        e->own_u32 = true;
        he = hipMemset(e->u32, 0, e->L.n_u32 * sizeof(uint32_t));
        if (he != hipSuccess) goto nomem;
    }
    if (e->p.accum == ITX_ACCUM_PARTITION) {
        int rc = itx_part_create(t, batch_capacity, &e->pw);
        if (rc != ITX_OK) {
            itx_engine_destroy(e);
            return rc;
        }
    }
    *out = e;
    return ITX_OK;
nomem:
    itx_set_error("itx_engine_create: device allocation failed: %s", hipGetErrorString(he));
    itx_engine_destroy(e);
    return ITX_E_NOMEM;
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
    if (e->slots_ready && e->compute) (void)hipStreamDestroy(e->compute);
    if (e->own_u64 && e->u64) (void)hipFree(e->u64);
    if (e->own_u32 && e->u32) (void)hipFree(e->u32);
    if (e->d_tidmap) (void)hipFree(e->d_tidmap);
    if (e->d_rep_out) (void)hipFree(e->d_rep_out);
    if (e->d_cov) (void)hipFree(e->d_cov);
    if (e->d_cov_uniq) (void)hipFree(e->d_cov_uniq);
    if (e->d_locus_out) (void)hipFree(e->d_locus_out);
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
        if (e->d_tidmap) ITX_HIP(hipFree(e->d_tidmap));
        e->d_tidmap = nullptr;
        ITX_HIP(hipMalloc((void **)&e->d_tidmap, sizeof(int32_t) * (size_t)(n_tid + 1)));
        e->cap_tid = n_tid;
    }
    if (n_tid) ITX_HIP(hipMemcpy(e->d_tidmap, tid2chrom, sizeof(int32_t) * (size_t)n_tid, hipMemcpyHostToDevice));
    e->n_tid = n_tid;
    return ITX_OK;
}

static int ensure_slots(itx_engine *e)
{
    if (e->slots_ready) return ITX_OK;
    const size_t n = e->cap;
    ITX_HIP(hipStreamCreateWithFlags(&e->compute, hipStreamNonBlocking));
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
        S.h.capacity = S.d.capacity = n;
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

static int run_batch(itx_engine *e, const ItxDevBatch &B, size_t n, int32_t *d_hit_row, hipStream_t st, bool accumulate)
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
    P.extension = e->p.extension;
    P.isize_max = e->p.isize_max;
    P.treat = e->p.treat_pe_as_se;
    P.discard = e->p.discard_half_mapped;
    P.mode = e->p.mode;
    P.n_tid = e->n_tid;
    P.tid2chrom = e->d_tidmap;
    hipEvent_t a, b;
    ITX_HIP(hipEventCreate(&a));
    ITX_HIP(hipEventCreate(&b));
    ITX_HIP(hipEventRecord(a, st));
    int rc;
    if (!accumulate)
        rc = itx_launch_atomic(e->t->dev, P, B, n, 0, d_hit_row, e->u64, e->u32, e->L, st);
    else if (e->p.accum == ITX_ACCUM_PARTITION)
        rc = itx_part_run(e->pw, e->t->dev, P, B, n, d_hit_row, e->u64, e->u32, e->L, st);
    else
        rc = itx_launch_atomic(e->t->dev, P, B, n, 1, d_hit_row, e->u64, e->u32, e->L, st);
    ITX_HIP(hipEventRecord(b, st));
    e->ev.emplace_back(a, b);
    e->records += n;
    return rc;
}

extern "C" int itx_engine_submit_slot(itx_engine *e, int slot, size_t n, int has_paired, int want_hits)
{
    if (!e || slot < 0 || slot > 1) {
        itx_set_error("itx_engine_submit_slot: bad argument");
        return ITX_E_ARG;
    }
    int rc = use_device(e);
    if (rc) return rc;
    rc = ensure_slots(e);
    if (rc) return rc;
    if (n > e->cap) {
        itx_set_error("itx_engine_submit_slot: %zu records exceed the slot capacity %zu", n, e->cap);
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
    rc = run_batch(e, B, n, want_hits ? S.d.hit_row : nullptr, e->compute, true);
    if (rc) return rc;
    ITX_HIP(hipEventRecord(S.done, e->compute));
    ITX_HIP(hipStreamWaitEvent(S.stream, S.done, 0));
    if (want_hits)
        ITX_HIP(hipMemcpyAsync(S.h.hit_row, S.d.hit_row, n * sizeof(int32_t), hipMemcpyDeviceToHost, S.stream));
    S.busy = true;
    return ITX_OK;
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
    return run_batch(e, to_dev_batch(b), n, d_hit_row, (hipStream_t)stream, true);
}

extern "C" int itx_engine_classify_device(itx_engine *e, const itx_batch *b, size_t n, int32_t *d_hit_row, void *stream)
{
    if (!e || !b || !d_hit_row || (n && (!b->tid || !b->pos || !b->tmpend || !b->mapq || !b->flag5))) {
        itx_set_error("itx_engine_classify_device: bad argument");
        return ITX_E_ARG;
    }
    int rc = use_device(e);
    if (rc) return rc;
    uint64_t rec0 = e->records;
    rc = run_batch(e, to_dev_batch(b), n, d_hit_row, (hipStream_t)stream, false);
    e->records = rec0;
    if (rc) return rc;
    return ITX_OK;
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
}

extern "C" int itx_engine_reset(itx_engine *e)
{
    if (!e) return ITX_E_ARG;
    int rc = itx_engine_sync(e);
    if (rc) return rc;
    ITX_HIP(hipMemset(e->u64, 0, e->L.n_u64 * sizeof(uint64_t)));
    ITX_HIP(hipMemset(e->u32, 0, e->L.n_u32 * sizeof(uint32_t)));
    fold_events(e);
    e->kernel_ms = 0;
    e->records = 0;
    return ITX_OK;
}

// locus counts are accumulated per SORTED row; the caller wants its own row order.
__global__ void k_permute_locus(const uint32_t *__restrict__ in, const int32_t *__restrict__ orig, uint32_t n,
                                uint32_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[orig[i]] = in[i];
}

extern "C" int itx_engine_finish(itx_engine *e, const itx_result *out)
{
    if (!e || !out) {
        itx_set_error("itx_engine_finish: bad argument");
        return ITX_E_ARG;
    }
    int rc = itx_engine_sync(e);
    if (rc) return rc;
    const itx_table *t = e->t;
    if (!e->d_rep_out) {
        ITX_HIP(hipMalloc((void **)&e->d_rep_out, sizeof(uint64_t) * (2 * (size_t)t->n_rep + 1)));
        ITX_HIP(hipMalloc((void **)&e->d_cov, sizeof(uint32_t) * (t->cov_len + 1)));
        ITX_HIP(hipMalloc((void **)&e->d_cov_uniq, sizeof(uint32_t) * (t->cov_len + 1)));
        ITX_HIP(hipMalloc((void **)&e->d_locus_out, sizeof(uint32_t) * ((size_t)t->n_rows + 1)));
    }
    if (out->cnt) {
        uint64_t c[16];
        ITX_HIP(hipMemcpy(c, e->u64 + e->L.cnt, sizeof c, hipMemcpyDeviceToHost));
        memcpy(out->cnt, c, 13 * sizeof(uint64_t));
    }
    if (e->p.mode == ITX_MODE_STAT) {
        rc = itx_launch_finalize(t, e->u64, e->u32, e->L, e->d_rep_out, e->d_cov, e->d_cov_uniq, 0);
        if (rc) return rc;
        ITX_HIP(hipDeviceSynchronize());
        if (out->rep_cnt && t->n_rep)
            ITX_HIP(hipMemcpy(out->rep_cnt, e->d_rep_out, sizeof(uint64_t) * 2 * (size_t)t->n_rep, hipMemcpyDeviceToHost));
        if (out->fam_cnt && t->n_fam)
            ITX_HIP(hipMemcpy(out->fam_cnt, e->u64 + e->L.fam, sizeof(uint64_t) * 2 * (size_t)t->n_fam, hipMemcpyDeviceToHost));
        if (out->cla_cnt && t->n_cla)
            ITX_HIP(hipMemcpy(out->cla_cnt, e->u64 + e->L.cla, sizeof(uint64_t) * 2 * (size_t)t->n_cla, hipMemcpyDeviceToHost));
        if (out->cov && t->cov_len) ITX_HIP(hipMemcpy(out->cov, e->d_cov, sizeof(uint32_t) * t->cov_len, hipMemcpyDeviceToHost));
        if (out->cov_uniq && t->cov_len)
            ITX_HIP(hipMemcpy(out->cov_uniq, e->d_cov_uniq, sizeof(uint32_t) * t->cov_len, hipMemcpyDeviceToHost));
        if (out->locus_cnt && t->n_rows) memset(out->locus_cnt, 0, sizeof(uint32_t) * (size_t)t->n_rows);
    } else {
        if (out->locus_cnt && t->n_rows) {
            hipLaunchKernelGGL(k_permute_locus, dim3((t->n_rows + 255) / 256), dim3(256), 0, 0, e->u32 + e->L.locus, t->dev.orig,
                               t->n_rows, e->d_locus_out);
            ITX_HIP(hipGetLastError());
            ITX_HIP(hipDeviceSynchronize());
            ITX_HIP(hipMemcpy(out->locus_cnt, e->d_locus_out, sizeof(uint32_t) * (size_t)t->n_rows, hipMemcpyDeviceToHost));
        }
        if (out->rep_cnt && t->n_rep) memset(out->rep_cnt, 0, sizeof(uint64_t) * 2 * (size_t)t->n_rep);
        if (out->fam_cnt && t->n_fam) memset(out->fam_cnt, 0, sizeof(uint64_t) * 2 * (size_t)t->n_fam);
        if (out->cla_cnt && t->n_cla) memset(out->cla_cnt, 0, sizeof(uint64_t) * 2 * (size_t)t->n_cla);
        if (out->cov && t->cov_len) memset(out->cov, 0, sizeof(uint32_t) * t->cov_len);
        if (out->cov_uniq && t->cov_len) memset(out->cov_uniq, 0, sizeof(uint32_t) * t->cov_len);
    }
    return ITX_OK;
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
    uint64_t c[16];
    ITX_HIP(hipMemcpy(c, e->u64 + e->L.cnt, sizeof c, hipMemcpyDeviceToHost));
    out->hits = c[9];
    return ITX_OK;
}
