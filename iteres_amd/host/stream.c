/* stream.c — the record loop of the reference (generic.c:700-1062 stat copy, 343-697 filter copy) as a host
 * pipeline around the GPU engine: decode a batch into one pinned staging slot while the other slot is in
 * flight (itx_engine_staging / submit_slot / wait_slot), keep what is order-dependent or text on the host
 * (progress banners, the "chromosome not in the size file" warnings, read names for filter -r, and — side.c —
 * the -R duplicate filter, the -B/-V bed lines and the XA veto, which reach the engine as one flag bit per record). */
#define _GNU_SOURCE
#include "itx_host.h"

#include <errno.h>
#include <stdlib.h>
#include <sys/stat.h>
#include <unistd.h>
#include <string.h>
#include <strings.h>
#include <time.h>

#define BATCH_RECORDS (4u << 20)

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* The first HIP call costs a few hundred milliseconds of runtime start-up; a helper thread pays them while the main
 * thread parses the rmsk file. */
#include <pthread.h>
static pthread_t warm_thread;
static int warm_on, warm_bam;
static char *warm_first;                      /* the first alignment file, opened (and decoded ahead) by the helper thread */
static aln_reader *warm_reader;
static int warm_single;                       /* the argument names one file */
static size_t warm_lo, warm_hi = SIZE_MAX;    /* the share the helper thread opened it with */

/* BAM input is decoded on the device (include/iteres_amd.h: itx_bamwin_*: blocks inflated, records located and parsed
 * there) unless ITX_HOST_INFLATE is set. The reader's compressed-chunk buffers then have to be page-locked, which takes
 * its time: the helper thread gets them ready while the main thread parses the rmsk file, and hands them out from this
 * little pool. */
static itx_inflater *g_inflater;
#define POOL_N (ITX_BAMWIN_LANES + 2)
static struct { void *p; size_t cap; int used; } pool[POOL_N];
static pthread_mutex_t pool_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t pool_cv = PTHREAD_COND_INITIALIZER;
static int pool_pinning;                       /* buffers the pinning thread (pin_main) has yet to deliver */
static void *pool_alloc(size_t n)
{
    pthread_mutex_lock(&pool_mu);
    int best = -1;
    for (;;) {
        for (int i = 0; i < POOL_N; i++)
            if (pool[i].p && !pool[i].used && pool[i].cap >= n && (best < 0 || pool[i].cap < pool[best].cap)) best = i;
        if (best >= 0 || pool_pinning <= 0) break;
        pthread_cond_wait(&pool_cv, &pool_mu);                       /* one is being locked right now: it will be here sooner than one of our own */
    }
    if (best >= 0) pool[best].used = 1;
    pthread_mutex_unlock(&pool_mu);
    if (best >= 0) return pool[best].p;
    /* a fresh one joins the pool when there is a place: released, it stays locked (for the next file; unlocking 384 MB takes
     * 30 ms, and at the end of the run the process leaves without) */
    void *p = itx_pinned_alloc(n);
    if (p) {
        pthread_mutex_lock(&pool_mu);
        for (int i = 0; i < POOL_N; i++)
            if (!pool[i].p) {
                pool[i].p = p;
                pool[i].cap = n;
                pool[i].used = 1;
                break;
            }
        pthread_mutex_unlock(&pool_mu);
    }
    return p;
}
static void pool_release(void *p)
{
    pthread_mutex_lock(&pool_mu);
    for (int i = 0; i < POOL_N; i++)
        if (pool[i].p == p) {
            pool[i].used = 0;                                        /* stays locked for the next file */
            pthread_mutex_unlock(&pool_mu);
            return;
        }
    pthread_mutex_unlock(&pool_mu);
    itx_pinned_free(p);
}

/* what the helper thread reserved on the device for the decoder (itx_inflater_reserve) */
static int g_dev_windows;
static size_t g_dev_max_blocks, g_dev_max_bytes;

static void use_device_reader(void)
{
    const aln_device_ops ops = {g_inflater,       itx_bamwin_push_begin, itx_bamwin_push_end, itx_bamwin_patch, itx_bamwin_truncate, itx_bamwin_carry, itx_bamwin_avail, itx_bamwin_peek,
                                itx_bamwin_skip, itx_bamwin_parse, itx_bamwin_fetch, itx_bamwin_bytes,    itx_bamwin_tids,  itx_bamwin_device_batch,
                                pool_alloc,      pool_release,     itx_last_error,   g_dev_windows,       g_dev_max_blocks, g_dev_max_bytes, itx_bamwin_xa_veto, itx_bamwin_push_copied};
    aln_use_device(&ops);
}

/* the shares of a multi-GPU job: shares.c (share_t, plan_shares) */
static int warm_splittable;
static size_t warm_input_bytes;               /* size of all alignment files together (0: unknown) */

/* The communicator of a multi-rank job is made while the stream is being read: ncclCommInitRank takes seconds (bootstrap
 * over the network interface, one ring per link) and needs nothing of the data — the exchange at the end only joins it. */
static struct {
    pthread_t th;
    int on, rc, done;
    itx_comm *comm;
    char err[400];
} early_comm;
static char warm_err[400];                    /* the helper thread's last error (itx_last_error is per thread) */
/* Which way the partials travel is agreed on by ALL ranks: every rank leaves a marker next to the communicator id when its
 * attempt at an RCCL communicator has ended — "ok" or "fail" — and the exchange is RCCL only when every marker says ok. A
 * rank that fails alone (its device, its copy of the library, the id file) would otherwise switch to files while the others
 * sit in ncclCommInitRank / ncclReduce, which have no timeout. */
static void comm_marker_path(char *buf, size_t n, int rank) { snprintf(buf, n, "%s.st%d", multi_comm_id(), rank); }
static void comm_marker_write(int ok)
{
    if (multi_world() <= 1) return;
    char path[700], tmp[720];
    comm_marker_path(path, sizeof path, multi_rank());
    snprintf(tmp, sizeof tmp, "%s.tmp", path);
    FILE *f = fopen(tmp, "w");
    if (!f) return;
    fputs(ok ? "ok" : "fail", f);
    if (fclose(f) == 0 && rename(tmp, path) != 0) unlink(tmp);
}
/* 1: every rank has a communicator; 0: some rank has none (all take the files); -1: a marker never came */
static int comm_agree(double timeout_s)
{
    const double t0 = now_s();
    for (unsigned spins = 0;; spins++) {
        int n_ok = 0;
        for (int r = 0; r < multi_world(); r++) {
            char path[700], w[8] = {0};
            comm_marker_path(path, sizeof path, r);
            FILE *f = fopen(path, "r");
            if (!f) continue;
            const size_t k = fread(w, 1, 7, f);
            fclose(f);
            if (k >= 4 && memcmp(w, "fail", 4) == 0) return 0;
            if (k >= 2 && memcmp(w, "ok", 2) == 0) n_ok++;
        }
        if (n_ok == multi_world()) return 1;
        if (now_s() - t0 > timeout_s) return -1;
        usleep(spins < 2000 ? 200 : 2000);
    }
}
static void comm_markers_remove(void)
{
    for (int r = 0; r < multi_world(); r++) {
        char path[700];
        comm_marker_path(path, sizeof path, r);
        unlink(path);
    }
}
static void *early_comm_main(void *arg)
{
    (void)arg;
    early_comm.rc = itx_comm_create(multi_rank(), multi_world(), multi_device(), multi_comm_id(), ITX_COMM_RCCL, &early_comm.comm);
    if (early_comm.rc != ITX_OK) snprintf(early_comm.err, sizeof early_comm.err, "%s", itx_last_error());
    comm_marker_write(early_comm.rc == ITX_OK);
    __atomic_store_n(&early_comm.done, 1, __ATOMIC_RELEASE);
    return NULL;
}

/* generic.c:781-801 for one reference name of a BAM header: its chromosome in the size file, -1 when it is not there (or its
 * size reads as "not found": generic.c:796-797), -2 when -C drops it */
static const char *rename_chr(const char *name, int add_chr, char *buf, size_t bufsz);
static int32_t chrom_of_target(const char *target, const run_opts *o, const sizes_t *chr_sizes)
{
    char buf[4096];
    const char *nm = rename_chr(target, o->add_chr, buf, sizeof buf);
    if (!nm) return -2;
    const int64_t id = names_find(&chr_sizes->names, nm);
    return (id >= 0 && (int)chr_sizes->value[id] != 2) ? (int32_t)id : -1;
}

/* ---- records parsed AHEAD of the table -------------------------------------------------------------------------------
 * The engine cannot take records before the table is on the device, and the ring of windows holds 0.1 s of decode: the decoder
 * used to sit idle for the rest of the rmsk parse and the table build. Now the helper thread goes on after it has opened the
 * first file: it parses the windows as they come, keeps their records in a backlog in HBM (include/iteres_amd.h itx_backlog_*:
 * 14 bytes per record instead of the 220 of the inflated stream) and lets the windows go back to the decoder. run_stream
 * submits the backlog first, in order. Only when nothing per record is the host's business (no -R, no bed files, no name
 * lists), one rank, and only whole windows without XA tags (the veto reads the window's bytes) and without mapped records on
 * chromosomes the size file lacks (their warnings are per record): the first window that does not qualify ends it and is
 * left, untouched, to the loop. ITX_NO_PREFETCH=1 turns it off. */
static const run_opts *pre_opts;
static int pre_filter_mode, pre_allowed;
static const sizes_t *pre_sizes;              /* set when the size file is loaded (stream_sizes_ready) */
static int pre_stop;                           /* run_stream: table and engine are there */
static itx_backlog *pre_bl;
static struct pre_ent { size_t at, n; int paired; } *pre_v;
static size_t pre_n, pre_cap;
static double pre_seconds;
void stream_prefetch_allow(const run_opts *o, int filter_mode, int allowed)
{
    pre_opts = o;
    pre_filter_mode = filter_mode;
    pre_allowed = allowed;
}
void stream_sizes_ready(const sizes_t *chr_sizes) { __atomic_store_n(&pre_sizes, chr_sizes, __ATOMIC_RELEASE); }

static void prefetch_records(aln_reader *rd)
{
    const double t0 = now_s();
    const sizes_t *cs = NULL;
    while (!(cs = __atomic_load_n(&pre_sizes, __ATOMIC_ACQUIRE))) {            /* (loaded milliseconds after this thread started) */
        if (__atomic_load_n(&pre_stop, __ATOMIC_ACQUIRE) || now_s() - t0 > 10) return;
        usleep(200);
    }
    const run_opts *o = pre_opts;
    const int nt = aln_n_targets(rd);
    if (nt <= 0) return;
    int32_t *t2c = xmalloc(sizeof(int32_t) * (size_t)nt);
    for (int t = 0; t < nt; t++) t2c[t] = chrom_of_target(aln_target_name(rd, t), o, cs);
    size_t cap = warm_input_bytes / 60 + ((size_t)1 << 20);                     /* no BAM record takes less than that compressed */
    if (cap > ((size_t)160 << 20)) cap = (size_t)160 << 20;
    if (itx_backlog_create(multi_device(), cap, &pre_bl) != ITX_OK) {
        pre_bl = NULL;
        free(t2c);
        return;
    }
    const int veto_on = o->xa_veto && !pre_filter_mode;
    while (!__atomic_load_n(&pre_stop, __ATOMIC_ACQUIRE)) {
        int wfl = 0, ok = 1;
        const uint8_t *seen = NULL;
        if (!aln_device_window(rd, &wfl, &seen)) break;                         /* end of input */
        if (veto_on && (wfl & 2)) break;
        for (int t = 0; t < nt && ok; t++)
            if (seen[t] && t2c[t] == -1) ok = 0;
        if (!ok || aln_device_left(rd) > itx_backlog_room(pre_bl)) break;
        itx_batch db;
        const size_t n = aln_read_batch_device(rd, SIZE_MAX, &db);
        if (n == 0) break;
        size_t at = 0;
        if (itx_backlog_append(pre_bl, &db, n, &at) != ITX_OK) die("records parsed ahead: %s", itx_last_error());
        if (pre_n == pre_cap) {
            pre_cap = pre_cap ? pre_cap * 2 : 64;
            pre_v = xrealloc(pre_v, sizeof *pre_v * pre_cap);
        }
        pre_v[pre_n].at = at;
        pre_v[pre_n].n = n;
        pre_v[pre_n].paired = db.mpos != NULL;
        pre_n++;
    }
    free(t2c);
    pre_seconds = now_s() - t0;
}

/* Page-locks the reader's chunk buffers, one after the other, from the moment the HIP runtime is up (0.1 s each for 384 MB):
 * the reader thread used to lock the third to fifth itself when it first needed them — 0.3 s during which the pipeline of
 * pushes crawled (one window decoded in the first 0.34 s). */
static size_t pin_want;
static int pin_count;
static void *pin_main(void *arg)
{
    (void)arg;
    for (int i = 0; i < pin_count && i < POOL_N; i++) {
        void *p = itx_pinned_alloc(pin_want);
        pthread_mutex_lock(&pool_mu);
        pool[i].p = p;
        pool[i].cap = p ? pin_want : 0;
        pool[i].used = 0;
        pool_pinning--;
        pthread_cond_broadcast(&pool_cv);
        pthread_mutex_unlock(&pool_mu);
    }
    pthread_mutex_lock(&pool_mu);
    pool_pinning = 0;
    pthread_cond_broadcast(&pool_cv);
    pthread_mutex_unlock(&pool_mu);
    return NULL;
}

static void *warm_main(void *arg)
{
    (void)arg;
    const double a = now_s();
    const int ndev = itx_device_count();
    const double b = now_s();
    if (ndev > 0 && (multi_world() > 1 || multi_selftest()) && multi_comm_mode() == ITX_COMM_RCCL && !getenv("ITX_NO_EARLY_COMM") &&
        pthread_create(&early_comm.th, NULL, early_comm_main, NULL) == 0)
        early_comm.on = 1;
    double t_created = b, t_pinned = b;
    int inf_rc = ITX_OK;
    /* the compressed chunks the reader rotates through are page-locked, which takes its time (0.1 - 0.2 s for the first two): a
     * thread of its own locks them while this one makes the inflater's streams; the others are locked by the reader thread when
     * it first needs them, beside the device's work */
    const char *ce = getenv("ITX_BGZF_CHUNK");
    const size_t chunk = ce && atol(ce) >= 1 ? (size_t)atol(ce) : ALN_DEVICE_CHUNK;
    size_t step = chunk;
    pthread_t pin_th;
    int pin_on = 0;
    if (ndev > 0 && warm_bam && !getenv("ITX_HOST_INFLATE")) {
        use_device_reader();                                             /* (aln_raw_step asks which decoder is in use; the inflater itself follows) */
        /* what the reader will ask for: a step's bytes plus room for a carried-over block; a small input gets small buffers */
        const size_t mine_bytes = warm_input_bytes ? warm_input_bytes / (size_t)(multi_world() > 0 ? multi_world() : 1) : 0;
        step = mine_bytes && !ce ? aln_raw_step(mine_bytes + mine_bytes / 64) : chunk;
        pin_want = step + (1u << 17);
        /* as many as a file of this size makes the reader use */
        pin_count = warm_input_bytes ? (int)(mine_bytes / step + 1) : 2;
        if (pin_count > ALN_DEVICE_RAW_BUFFERS) pin_count = ALN_DEVICE_RAW_BUFFERS;
        if (getenv("ITX_PIN_EARLY") && atoi(getenv("ITX_PIN_EARLY")) >= 1 && atoi(getenv("ITX_PIN_EARLY")) < pin_count) pin_count = atoi(getenv("ITX_PIN_EARLY"));   /* (A/B) */
        pool_pinning = pin_count;
        pin_on = pthread_create(&pin_th, NULL, pin_main, NULL) == 0;
        if (!pin_on) pool_pinning = 0;
    }
    if (ndev > 0 && warm_bam && !getenv("ITX_HOST_INFLATE") && (inf_rc = itx_inflater_create(multi_device(), &g_inflater)) != ITX_OK)
        snprintf(warm_err, sizeof warm_err, "%s", itx_last_error());
    else if (ndev <= 0)
        snprintf(warm_err, sizeof warm_err, "no usable GPU (%s)", itx_last_error());
    t_created = now_s();
    if (!pin_on && g_inflater) {
        pin_count = 2;
        pool_pinning = 2;
        pin_main(NULL);
    }
    t_pinned = now_s();
    if (!g_inflater) aln_use_device(NULL);                              /* it did not come up: the host decodes (run_stream says why when that will not do) */
    if (g_inflater) {
        /* the device side of the decoder, all of it, now that nothing runs there yet: windows and per-push scratch sized for
         * the most a push may carry (the block indexer cuts a chunk that inflates to more into two pushes) */
        const char *be = getenv("ITX_DEV_WINDOW_BLOCKS");
        const char *me = getenv("ITX_RESERVE_BLOCKS");                   /* scratch experiments */
        size_t max_blocks = be && atol(be) >= 1 ? (size_t)atol(be) : me && atol(me) >= 64 ? (size_t)atol(me) : 12288;
        /* no more than this rank's share of the input can need: a small file gets small buffers (a block of a real BAM takes
         * kilobytes; a file of smaller ones is simply cut into more pushes by the indexer) */
        if (warm_input_bytes) {
            const size_t mine = warm_input_bytes / (size_t)(multi_world() > 0 ? multi_world() : 1);
            const size_t est = mine / 2048 + 64;
            if (!be && est < max_blocks) max_blocks = est;
            const size_t nchunks = mine / chunk + 3;
            if (nchunks < ITX_BAMWIN_WINDOWS) g_dev_windows = (int)nchunks;          /* in: the most this input can use */
        }
        /* The ring the pushes rotate through: the pushes in flight plus four windows for the consumer. A deeper ring
         * (round 2: up to 48 windows, 50 GB of HBM for a 23 GB file) only let the decoder run ahead while the table was still
         * being built, and paid for it in device allocation time — seconds on a box whose memory the driver had yet to clear;
         * measured on 200 M reads: 8 windows 2.27 - 2.35 s per run, 48 windows 2.37 - 2.58 s. ITX_RESERVE_WINDOWS overrides. */
        {
            const char *we = getenv("ITX_RESERVE_WINDOWS");
            const char *pe = getenv("ITX_PUSHES");
            const int lanes = pe && atoi(pe) >= 1 && atoi(pe) <= ITX_BAMWIN_LANES ? atoi(pe) : ITX_BAMWIN_LANES_DEFAULT;
            const int ring = we && atol(we) >= 2 && atol(we) <= ITX_BAMWIN_WINDOWS ? (int)atol(we) : lanes + 4;
            if (g_dev_windows < 1 || g_dev_windows > ring) g_dev_windows = ring;
        }
        const size_t max_bytes = max_blocks * 65280u < ((size_t)1 << 30) ? max_blocks * 65280u : (size_t)1 << 30;
        if (!getenv("ITX_NO_RESERVE") && itx_inflater_reserve(g_inflater, step + (1u << 17), max_blocks, max_bytes, &g_dev_windows) == ITX_OK) {
            g_dev_max_blocks = max_blocks;
            g_dev_max_bytes = max_bytes;
        } else {
            g_dev_windows = 0;
        }
    }
    const double c = now_s();
    if (g_inflater && warm_first) {
        /* the first file's header, and its first windows decoded while the main thread is still parsing the rmsk file; a
         * file that does not open is left to run_stream, which reports it where the reference does */
        use_device_reader();
        share_t sh0;
        char *one[1] = {warm_first};
        /* this rank's share of the FIRST file when the list has one file (the common case); longer lists are opened by the loop */
        if (multi_world() <= 1) {
            warm_reader = aln_open(warm_first, 0);
        } else if (warm_single && plan_shares(one, 1, warm_splittable, multi_rank(), multi_world(), multi_min_share(), &sh0) && sh0.lo != sh0.hi) {
            warm_reader = aln_open_range(warm_first, sh0.lo, sh0.hi);
            warm_lo = sh0.lo;
            warm_hi = sh0.hi;
        }
        if (warm_reader) aln_readahead(warm_reader);
    }
    const double t_opened = now_s();
    if (getenv("ITX_TIMING"))
        fprintf(stderr, "[itx timing] HIP runtime start-up %.3f s, device inflater + page-locked buffers %.3f s (streams %.3f, page-locked %.3f, device reserve %.3f), first file opened %.3f s (helper thread)\n", b - a,
                c - b, t_created - b, t_pinned - t_created, c - t_pinned, t_opened - c);
    {
        const char *me = getenv("ITX_PREFETCH_MIN");                       /* bytes of input below which it is not worth a backlog (tests: 0) */
        const size_t min_bytes = me ? (size_t)atoll(me) : (size_t)1 << 30;
        if (warm_reader && pre_allowed && pre_opts && multi_world() <= 1 && warm_input_bytes >= min_bytes && !getenv("ITX_NO_PREFETCH")) prefetch_records(warm_reader);
    }
    if (pin_on) pthread_join(pin_th, NULL);                              /* (long done: it started when the runtime came up) */
    return NULL;
}
void gpu_warmup_start(int bam_input, const char *aln_arg, int multi_file, int splittable)
{
    warm_bam = bam_input;
    warm_splittable = splittable;
    if (bam_input && aln_arg) {
        warm_first = xstrdup(aln_arg);
        char *c = multi_file ? strchr(warm_first, ',') : NULL;
        warm_single = c == NULL;
        if (c) *c = 0;
        char *all = xstrdup(aln_arg), *save = NULL;
        warm_input_bytes = 0;
        for (char *tok = multi_file ? strtok_r(all, ",", &save) : all; tok; tok = multi_file ? strtok_r(NULL, ",", &save) : NULL) {
            struct stat sb;
            if (stat(tok, &sb) == 0 && S_ISREG(sb.st_mode)) warm_input_bytes += (size_t)sb.st_size;
            else {
                warm_input_bytes = 0;                                   /* a pipe: no idea */
                break;
            }
        }
        free(all);
    }
    if (!warm_on && pthread_create(&warm_thread, NULL, warm_main, NULL) == 0) warm_on = 1;
}
static void gpu_warmup_join(void)
{
    if (warm_on) pthread_join(warm_thread, NULL);
    warm_on = 0;
}

static void chk(int rc, const char *what)
{
    if (rc != ITX_OK) die("%s: %s", what, itx_last_error());
}

/* -R on the device: every window, right after its records are parsed (aln_set_window_hook) */
static double t_dedup;
static void dedup_window(void *ctx, size_t n_rec)
{
    (void)n_rec;
    const double t0 = now_s();
    chk(itx_bamwin_dedup(g_inflater, (itx_dedup *)ctx), "itx_bamwin_dedup");
    t_dedup += now_s() - t0;
}

/* generic.c:781-791 (-C): NULL when the record is skipped ("GL*"), else the (possibly renamed) chromosome */
static const char *rename_chr(const char *name, int add_chr, char *buf, size_t bufsz)
{
    if (!add_chr) return name;
    if (strncmp(name, "GL", 2) == 0) return NULL;
    if (strcasecmp(name, "MT") == 0) return "chrM";
    if (strncmp(name, "chr", 3) != 0) {
        snprintf(buf, bufsz, "chr%s", name);
        return buf;
    }
    return name;
}

typedef struct {
    uint32_t row;
    char *name;
} hit_name;

typedef struct {
    hit_name *v;
    size_t n, cap;
} hit_names;

/* filter -r: keep the names of the records that chose a row, free the others */
static void collect_names(hit_names *hn, const itx_staging *st, aln_side *side, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        const int32_t row = st->hit_row[i];
        if (row >= 0) {
            if (hn->n == hn->cap) {
                hn->cap = hn->cap ? hn->cap * 2 : 1 << 16;
                hn->v = xrealloc(hn->v, sizeof *hn->v * hn->cap);
            }
            hn->v[hn->n].row = (uint32_t)row;
            hn->v[hn->n].name = side->qname[i];
            hn->n++;
        } else {
            free(side->qname[i]);
        }
        side->qname[i] = NULL;
    }
}

static void side_release(aln_side *side, size_t n, int keep_qnames)
{
    if (!side->has_strings) return;                /* nothing was stored: the arrays are all NULL as they were */
    side->has_strings = 0;
    for (size_t i = 0; i < n; i++) {
        if (side->want_qnames && !keep_qnames) {
            free(side->qname[i]);
            side->qname[i] = NULL;
        }
        if (side->want_aux) {
            free(side->xa[i]);
            side->xa[i] = NULL;
        }
    }
}

/* after a multi-GPU stream: the reduced partial (the engine's own buffers) the writers' arrays come from */
static void *g_reduced_u64, *g_reduced_u32;

int stream_finish(itx_engine *eng, const itx_result *res)
{
    if (g_reduced_u64) return itx_engine_finish_partial(eng, g_reduced_u64, g_reduced_u32, res);
    return itx_engine_finish(eng, res);
}

void run_stream(const run_opts *o, const rmsk_t *rm, const sizes_t *chr_sizes, int filter_mode, int multi_file,
                unsigned progress_every, int want_qnames, itx_engine **eng_out, itx_table **tab_out, char ***locus_names,
                host_counts *hc)
{
    const int timing = getenv("ITX_TIMING") != NULL;
    struct timespec ts0, ts1;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    int ndev = itx_device_count();
    if (ndev <= 0) die("no usable MI355X (HIP) device: %s", ndev < 0 ? itx_last_error() : "none visible");
    /* table: every chromosome of the size file is known to the engine (a read may land on one without repeats) */
    const uint32_t n_chrom = chr_sizes->names.n;
    itx_table *tab = NULL;
    size_t bad = 0;
    int rc = itx_table_create(rm->rows, rm->n_rows, chr_sizes->value, (int)n_chrom, rm->rep_len, rm->reps.n, rm->fams.n, rm->clas.n, multi_device(),
                              &tab, &bad);
    if (rc == ITX_E_RANGE) {
        const itx_row *r = &rm->rows[bad];
        die("(%d %d) out of range (%d %d) in binKeeperAdd", (int)r->start, (int)r->end, 0, (int)chr_sizes->value[r->chrom]);
    }
    chk(rc, "itx_table_create");
    itx_params p;
    memset(&p, 0, sizeof p);
    p.mapq_min = o->mapq;
    p.min_cov = o->min_cov;
    p.extension = o->extension;
    p.isize_max = o->isize;
    p.treat_pe_as_se = o->treat;
    p.discard_half_mapped = o->discard;
    p.mode = filter_mode ? ITX_MODE_FILTER : ITX_MODE_STAT;
    p.accum = ITX_ACCUM_DEFAULT;
    itx_engine *eng = NULL;
    chk(itx_engine_create(tab, &p, BATCH_RECORDS, &eng), "itx_engine_create");
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    if (timing) fprintf(stderr, "[itx timing] table build + engine %.3f s\n", (double)(ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts0.tv_nsec));
    itx_staging st[2];                            /* the pinned slots of the host route, taken when a batch first goes that way */
    int have_slots = 0;
    memset(st, 0, sizeof st);

    /* what the host keeps per record beside the SoA */
    const int want_bed = o->bed_path || o->bed_uniq_path;
    const int veto_on = o->xa_veto && !filter_mode;                    /* filter.c:134 passes diffSubfam = 0 */
    aln_side side[2];
    size_t pend[2] = {0, 0};
    for (int k = 0; k < 2; k++) {
        memset(&side[k], 0, sizeof side[k]);
        side[k].want_qnames = want_qnames || want_bed;
        side[k].want_aux = veto_on || o->bed_path != NULL;
        if (side[k].want_qnames) side[k].qname = xcalloc(BATCH_RECORDS, sizeof(char *));
        if (side[k].want_aux) {
            side[k].xa = xcalloc(BATCH_RECORDS, sizeof(char *));
            side[k].nm = xcalloc(BATCH_RECORDS, sizeof(int32_t));
        }
    }
    const int any_side = side[0].want_qnames || side[0].want_aux;
    hit_names hn = {NULL, 0, 0};
    host_iv *iv = NULL;
    uint8_t *live = NULL;                                              /* the record reached the bed / veto stage */
    if (o->dedup || want_bed || veto_on) {
        iv = xcalloc(BATCH_RECORDS, sizeof *iv);
        live = xcalloc(BATCH_RECORDS, 1);
    }
    itx_dedup *dd = NULL;
    dup_set *dups = NULL;
    names_t chr_names;                                                 /* identities of the chromosome strings inside -R keys */
    names_init(&chr_names);
    xa_index *xi = NULL;
    itx_xaveto *xv = NULL;                                             /* the veto on the device (windows that stay in HBM) */
    const int dev_veto = !getenv("ITX_HOST_VETO");
    unsigned long long veto_dev_batches = 0, veto_host_batches = 0;
    FILE *bed_f = NULL, *bed_uniq_f = NULL;
    /* mustOpen, cuskent/common.c:2543-2568 */
    if (o->bed_path && !(bed_f = fopen(o->bed_path, "w"))) die("mustOpen: Can't open %s to write: %s", o->bed_path, strerror(errno));
    if (o->bed_uniq_path && !(bed_uniq_f = fopen(o->bed_uniq_path, "w")))
        die("mustOpen: Can't open %s to write: %s", o->bed_uniq_path, strerror(errno));
    unsigned long long ends = 0;

    /* the list of files (stat: comma separated, generic.c:725; filter: one file) */
    char *arg = xstrdup(o->aln_arg);
    char *files[100];
    int n_files = 0;
    if (multi_file) {
        for (char *s = arg; n_files < 100;) {
            files[n_files++] = s;
            char *c = strchr(s, ',');
            if (!c) break;
            *c = 0;
            s = c + 1;
        }
    } else {
        files[n_files++] = arg;
    }
    names_t warned;
    names_init(&warned);
    /* this rank's share of every file (one rank: all of it) */
    share_t share[100];
    const int world = multi_world(), rank = multi_rank();
    int shared = plan_shares(files, n_files, warm_splittable && !getenv("ITX_HOST_INFLATE"), rank, world, multi_min_share(), share) && world > 1;
    unsigned long long boundary_missed = 0;
    /* the helper thread (HIP start-up, device decoder, first file opened and decoding ahead) has had the rmsk parse and the
     * table build to finish */
    const double t_join = now_s();
    if (getenv("ITX_PREFETCH_HOLD_MS")) usleep((useconds_t)atol(getenv("ITX_PREFETCH_HOLD_MS")) * 1000u);     /* tests: a table that takes its time */
    __atomic_store_n(&pre_stop, 1, __ATOMIC_RELEASE);                  /* table and engine are there: the helper thread finishes the window it is at */
    gpu_warmup_join();
    if (timing) fprintf(stderr, "[itx timing] waited %.3f s for the helper thread\n", now_s() - t_join);
    if (g_inflater) use_device_reader();
    /* -R: one set over all files, like `dup` (generic.c:721). BAM decoded on the device: the set lives there too
     * (csrc/itx_dedup.hip) and marks a window's duplicates where the window lies, right after it is parsed; SAM text, the host
     * decoder and ITX_HOST_DEDUP=1: the host's hash set, record by record */
    if (o->dedup && g_inflater && !o->is_sam && !getenv("ITX_HOST_DEDUP")) {
        size_t cells = warm_input_bytes / 24;                          /* a first guess at the keys to come; the table grows */
        if (cells < ((size_t)1 << 20)) cells = (size_t)1 << 20;
        if (cells > ((size_t)1 << 28)) cells = (size_t)1 << 28;
        chk(itx_dedup_create(multi_device(), chr_sizes->value, (int)n_chrom, &p, cells, &dd), "itx_dedup_create");
    }
    dups = o->dedup && !dd ? dup_set_new() : NULL;
    if (shared && !g_inflater) die("rank %d: the device decoder did not come up: %s", rank, warm_err[0] ? warm_err : itx_last_error());
    /* pass 0: this rank's shares, then the exchange. pass 1 (rank 0 only, and only when a share boundary did not hold —
     * the split points are guesses that the rank before verifies): the whole job again by this rank alone */
    for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) {
        if (!boundary_missed) break;
        fprintf(stderr, "[iteres] note: a share boundary was not a record start; scanning the input again with one GPU\n");
        chk(itx_engine_reset(eng), "itx_engine_reset");
        if (hc) hc->diff_subfam = hc->dup_unique = 0;
        ends = 0;
        for (int i = 0; i < n_files; i++) share[i].lo = 0, share[i].hi = SIZE_MAX;
        shared = 0;
    }
    for (int fi = 0; fi < n_files; fi++) {
        if (multi_file) fprintf(stderr, "\n* Processing %s\n", files[fi]);
        if (share[fi].lo == share[fi].hi) continue;                      /* nothing of this file is this rank's */
        const double t_open0 = now_s();
        aln_reader *rd = NULL;
        int drain_backlog = 0;
        if (fi == 0 && pass == 0 && warm_reader && warm_first && strcmp(files[0], warm_first) == 0 && warm_lo == share[0].lo && warm_hi == share[0].hi) {
            rd = warm_reader;                                            /* opened and decoding since the helper thread came up */
            warm_reader = NULL;
            drain_backlog = pre_bl != NULL;
        } else {
            if (warm_reader && fi == 0 && pass == 0) {                   /* opened for another share than this plan's: not used */
                aln_close(warm_reader);
                warm_reader = NULL;
            }
            rd = (share[fi].lo == 0 && share[fi].hi == SIZE_MAX) ? aln_open(files[fi], o->is_sam) : aln_open_range(files[fi], share[fi].lo, share[fi].hi);
        }
        const double t_opened = now_s();
        if (!rd) {
            fprintf(stderr, "Fail to open %s file %s\n", o->is_sam ? "SAM" : "BAM", o->aln_arg);
            die("Error\n");
        }
        const int nt = aln_n_targets(rd);
        int32_t *t2c = xmalloc(sizeof(int32_t) * (size_t)(nt + 1));
        uint32_t *t2id = xcalloc((size_t)nt + 1, sizeof(uint32_t));
        char **t2name = xcalloc((size_t)nt + 1, sizeof(char *));
        for (int t = 0; t < nt; t++) {
            char buf[4096];
            const char *nm = rename_chr(aln_target_name(rd, t), o->add_chr, buf, sizeof buf);
            t2name[t] = nm ? xstrdup(nm) : NULL;
            if (!nm) {
                t2c[t] = -2;
            } else {
                /* generic.c:796-797: cend = size-1 with 2 as the "not found" default; a listed size of 2 reads the same */
                t2c[t] = chrom_of_target(aln_target_name(rd, t), o, chr_sizes);
                if (dups || dd) t2id[t] = names_intern(&chr_names, nm);
            }
        }
        chk(itx_engine_set_tidmap(eng, t2c, nt > 0 ? nt : 0), "itx_engine_set_tidmap");
        if (xv) chk(itx_xaveto_set_tidmap(xv, t2c, nt > 0 ? nt : 0), "itx_xaveto_set_tidmap");
        if (dd) {
            chk(itx_dedup_set_tidmap(dd, t2c, t2id, nt > 0 ? nt : 0), "itx_dedup_set_tidmap");
            aln_set_window_hook(rd, dedup_window, dd);
        }
        if (nt == 0) {
            /* no references: nothing can map; still count the read ends */
            int32_t none = -1;
            chk(itx_engine_set_tidmap(eng, &none, 1), "itx_engine_set_tidmap");
        }
        int s = 0, any_paired = 0, aux_xa = 0;
        double t_wait = 0, t_read = 0, t_host = 0, t_submit = 0, tq;
        /* Nothing per record is the host's business when no option asks for names, -R or bed lines: a window of records the
         * device decoder parsed then goes to the engine where it lies, in HBM. What is left for the host to look at comes with
         * the window — does a record carry an XA tag (the veto needs its strings: host route for that window), is a mapped
         * record on a chromosome the size file lacks (the warning is per record, in file order: host route). */
        const int handoff_ok = !dups && !bed_f && !bed_uniq_f && !want_qnames;
        if (drain_backlog) {
            /* the records the helper thread parsed while the table was being built: first, in file order */
            const double td = now_s();
            unsigned long long got = 0;
            for (size_t k = 0; k < pre_n; k++)
                for (size_t off = 0; off < pre_v[k].n; off += BATCH_RECORDS) {
                    const size_t m = pre_v[k].n - off < BATCH_RECORDS ? pre_v[k].n - off : BATCH_RECORDS;
                    itx_batch db;
                    chk(itx_backlog_batch(pre_bl, pre_v[k].at + off, pre_v[k].paired, &db), "itx_backlog_batch");
                    for (unsigned long long mk = (ends / progress_every + 1) * progress_every; mk <= ends + m; mk += progress_every)
                        fprintf(stderr, "\r* Processed read ends: %llu", mk);
                    ends += m;
                    got += m;
                    chk(itx_engine_submit_device_own(eng, &db, m, NULL), "itx_engine_submit_device_own");
                }
            chk(itx_engine_wait_own(eng), "itx_engine_wait_own");
            if (timing)
                fprintf(stderr, "\n[itx timing] parsed ahead of the table by the helper thread: %llu records of %zu windows (%.3f s there), submitted in %.3f s\n", got, pre_n,
                        pre_seconds, now_s() - td);
        }
        if (pre_bl && fi == 0 && pass == 0) {                            /* used or not (another share than the plan's): gone */
            itx_backlog_destroy(pre_bl);
            pre_bl = NULL;
            free(pre_v);
            pre_v = NULL;
            pre_n = pre_cap = 0;
        }
        for (;;) {
            if (handoff_ok) {
                int wfl = 0, direct = 1;
                const uint8_t *seen = NULL;
                tq = now_s();
                if (aln_device_window(rd, &wfl, &seen)) {
                    /* a window with XA tags while the veto is on: the veto runs on the device too (itx_xaveto_*), batch by batch
                     * (a window of records without sequence holds several batches); when a record needs the host's reading, the
                     * host route takes over from that batch on */
                    const int xa_window = veto_on && (wfl & 2);
                    if (xa_window && !dev_veto) direct = 0;
                    for (int t = 0; t < nt && direct; t++)
                        if (seen[t] && t2c[t] == -1) direct = 0;
                    if (direct && xa_window && !xv) {
                        uint32_t *row_rep = xmalloc(sizeof(uint32_t) * (rm->n_rows + 1)), *words = xa_rep_words(rm);
                        for (size_t i = 0; i < rm->n_rows; i++) row_rep[i] = rm->rows[i].rep;
                        chk(itx_xaveto_create(tab, &p, row_rep, words, (const char *const *)chr_sizes->names.name, (int)n_chrom, BATCH_RECORDS, &xv), "itx_xaveto_create");
                        chk(itx_xaveto_set_tidmap(xv, t2c, nt > 0 ? nt : 0), "itx_xaveto_set_tidmap");
                        free(row_rep);
                        free(words);
                    }
                    if (direct) {
                        int wfl2;
                        do {
                            itx_batch db;
                            const size_t n = aln_read_batch_device(rd, BATCH_RECORDS, &db);
                            t_read += now_s() - tq;
                            if (n == 0) break;
                            if (xa_window) {
                                /* classify, let the device read the tags of the classified records, mark the vetoed ones */
                                const double tv = now_s();
                                uint64_t vetoed = 0, hard = 0;
                                chk(itx_engine_classify_device(eng, &db, n, itx_xaveto_hits(xv), itx_xaveto_stream(xv)), "itx_engine_classify_device");
                                if (aln_device_xa_veto(rd, xv, n, &vetoed, &hard) != 0) die("device veto: %s", itx_last_error());
                                t_host += now_s() - tv;
                                if (hard) {                                       /* an alternative only strtol / the reference's assert can judge: this batch
                                                                                   * and the rest of the window take the host route (the batches before are through) */
                                    aln_device_rewind(rd, n);
                                    direct = 0;
                                    tq = now_s();
                                    break;
                                }
                                if (hc) hc->diff_subfam += vetoed;
                                veto_dev_batches++;
                            }
                            for (unsigned long long m = (ends / progress_every + 1) * progress_every; m <= ends + n; m += progress_every)
                                fprintf(stderr, "\r* Processed read ends: %llu", m);
                            ends += n;
                            tq = now_s();
                            chk(itx_engine_submit_device_own(eng, &db, n, NULL), "itx_engine_submit_device_own");
                            /* the window's arrays are the decoder's again once its last batch is through: the next parse
                             * overwrites them */
                            if (aln_device_left(rd) == 0) chk(itx_engine_wait_own(eng), "itx_engine_wait_own");
                            t_submit += now_s() - tq;
                            tq = now_s();
                        } while (!aln_device_window(rd, &wfl2, &seen));                        /* until the next window's start (or the end) */
                        if (direct) continue;
                    }
                }
                t_read += now_s() - tq;
                if (aln_device_exhausted(rd)) break;
            }
            if (!have_slots) {
                chk(itx_engine_staging(eng, 0, &st[0]), "itx_engine_staging");
                chk(itx_engine_staging(eng, 1, &st[1]), "itx_engine_staging");
                have_slots = 1;
            }
            /* slot s: collect what its previous batch left behind, then refill */
            tq = now_s();
            chk(itx_engine_wait_slot(eng, s), "itx_engine_wait_slot");
            t_wait += now_s() - tq;
            if (pend[s]) {
                if (want_qnames) collect_names(&hn, &st[s], &side[s], pend[s]);
                side_release(&side[s], pend[s], want_qnames);
                pend[s] = 0;
            }
            aux_xa = 0;                                                        /* per batch */
            tq = now_s();
            const size_t n = aln_read_batch(rd, &st[s], BATCH_RECORDS, any_side ? &side[s] : NULL, &any_paired, &aux_xa);
            t_read += now_s() - tq;
            if (n == 0) break;
            tq = now_s();
            /* generic.c:760-761 progress; generic.c:793-801 one warning per unknown chromosome, in file order. Batches
             * without a mapped record on an unknown chromosome (all of them, for most files) only print the progress marks. */
            int unknown = 0;
            {
                const int32_t *tidv = st[s].tid;
                const uint8_t *f5 = st[s].flag5;
#pragma omp parallel for schedule(static) reduction(| : unknown)
                for (long i = 0; i < (long)n; i++) {
                    const int32_t t = tidv[i];
                    if (!(f5[i] & 2) && t >= 0 && t < nt && t2c[t] == -1) unknown |= 1;
                }
            }
            if (!unknown) {
                for (unsigned long long m = (ends / progress_every + 1) * progress_every; m <= ends + n; m += progress_every)
                    fprintf(stderr, "\r* Processed read ends: %llu", m);
                ends += n;
            } else {
                for (size_t i = 0; i < n; i++) {
                    if (++ends % progress_every == 0) fprintf(stderr, "\r* Processed read ends: %llu", ends);
                    const int32_t t = st[s].tid[i];
                    if (!(st[s].flag5[i] & 2) && t >= 0 && t < nt && t2c[t] == -1 && names_find(&warned, t2name[t]) < 0) {
                        names_intern(&warned, t2name[t]);
                        warnf("* Warning: read ends mapped to chromosome %s will be discarded as %s not existed in the chromosome size file",
                              t2name[t], t2name[t]);
                    }
                }
            }
            /* ---- in file order: -R and the bed lines (generic.c:907-936) */
            int batch_xa = 0;
            if (dups || dd || bed_f || bed_uniq_f) {
                for (size_t i = 0; i < n; i++) {
                    const int32_t t = st[s].tid[i];
                    const int32_t chrom = (t >= 0 && t < nt) ? t2c[t] : -1;
                    live[i] = 0;
                    if (!host_derive(o, chrom, chrom >= 0 ? chr_sizes->value[chrom] : 0, st[s].flag5[i], st[s].pos[i], st[s].tmpend[i],
                                     st[s].mpos[i], st[s].isize[i], &iv[i]))
                        continue;
                    const int uniq = st[s].mapq[i] >= o->mapq;
                    if (dd && (st[s].flag5[i] & ITX_F5_NOLOOKUP)) continue;          /* a duplicate, marked on the device */
                    if (dups && dup_set_seen(dups, t2id[t], &iv[i], uniq)) {
                        st[s].flag5[i] |= ITX_F5_NOLOOKUP;
                        if (uniq && hc) hc->dup_unique++;
                        continue;
                    }
                    live[i] = 1;
                    const char *xa = side[s].want_aux ? side[s].xa[i] : NULL;
                    if (xa) batch_xa = 1;
                    if (bed_f) {
                        fprintf(bed_f, "%s\t%u\t%u\t%s\t%i\t%c", t2name[t], iv[i].start, iv[i].end, side[s].qname[i], (int)st[s].mapq[i], iv[i].strand);
                        if (xa) fprintf(bed_f, "\t%i\t%s", (int)side[s].nm[i], xa);
                        fprintf(bed_f, "\n");
                    }
                    if (bed_uniq_f && uniq)
                        fprintf(bed_uniq_f, "%s\t%u\t%u\t%s\t%i\t%c\n", t2name[t], iv[i].start, iv[i].end, side[s].qname[i], (int)st[s].mapq[i],
                                iv[i].strand);
                }
            } else if (veto_on && aux_xa) {
                /* only the veto needs the intervals, and only of the records that carry XA: no order involved */
                batch_xa = 1;
#pragma omp parallel for schedule(static)
                for (long i = 0; i < (long)n; i++) {
                    live[i] = 0;
                    if (!side[s].xa[i] || (st[s].flag5[i] & ITX_F5_NOLOOKUP)) continue;
                    const int32_t t = st[s].tid[i];
                    const int32_t chrom = (t >= 0 && t < nt) ? t2c[t] : -1;
                    live[i] = (uint8_t)host_derive(o, chrom, chrom >= 0 ? chr_sizes->value[chrom] : 0, st[s].flag5[i], st[s].pos[i],
                                                   st[s].tmpend[i], st[s].mpos[i], st[s].isize[i], &iv[i]);
                }
            }
            /* ---- the XA veto needs the chosen row first: classify the slot, look, mark, then count (generic.c:972-982) */
            if (veto_on && batch_xa) {
                veto_host_batches++;
                if (!xi) xi = xa_index_new(rm);
                chk(itx_engine_classify_slot(eng, s, n, any_paired), "itx_engine_classify_slot");
                chk(itx_engine_wait_slot(eng, s), "itx_engine_wait_slot");
                unsigned long long vetoed = 0;
#pragma omp parallel for schedule(dynamic, 4096) reduction(+ : vetoed)
                for (long i = 0; i < (long)n; i++) {
                    if (!live[i] || !side[s].xa[i] || st[s].hit_row[i] < 0) continue;
                    const uint32_t rep = rm->rows[st[s].hit_row[i]].rep;
                    if (xa_veto(xi, rep, side[s].nm[i], side[s].xa[i], (int)(iv[i].end - iv[i].start))) {
                        st[s].flag5[i] |= ITX_F5_NOLOOKUP;
                        vetoed++;
                    }
                }
                if (hc) hc->diff_subfam += vetoed;
            }
            t_host += now_s() - tq;
            tq = now_s();
            chk(itx_engine_submit_slot(eng, s, n, any_paired, want_qnames), "itx_engine_submit_slot");
            t_submit += now_s() - tq;
            pend[s] = n;
            s ^= 1;
        }
        const double t_loop_done = now_s();
        /* drain both slots before the tid map of the next file replaces this one */
        for (int k = 0; k < 2; k++) {
            chk(itx_engine_wait_slot(eng, k), "itx_engine_wait_slot");
            if (pend[k]) {
                if (want_qnames) collect_names(&hn, &st[k], &side[k], pend[k]);
                side_release(&side[k], pend[k], want_qnames);
            }
            pend[k] = 0;
        }
        /* a share of a multi-rank job: the line with the whole job's count comes after the exchange (rank 0's share alone
         * would differ from what the reference prints) */
        if (!(shared && pass == 0)) fprintf(stderr, "\r* Processed read ends: %llu\n", ends);
        if (timing)
            fprintf(stderr, "[itx timing] stream of %s: decode %.3f s, host passes %.3f s, submit %.3f s, waiting for the device %.3f s\n", files[fi],
                    t_read, t_host, t_submit, t_wait);
        for (int t = 0; t < nt; t++) free(t2name[t]);
        free(t2name);
        free(t2id);
        free(t2c);
        const double t_drained = now_s();
        if (!aln_range_verified(rd)) boundary_missed++;
        aln_close(rd);
        if (timing)
            fprintf(stderr, "[itx timing] open %.3f s, record loop %.3f s, drain %.3f s, close %.3f s\n", t_opened - t_open0, t_loop_done - t_opened,
                    t_drained - t_loop_done, now_s() - t_drained);
    }
    if (pass == 1 || (world <= 1 && !multi_selftest())) break;
    {
        /* ---- the ONE exchange: every rank's partial, summed onto rank 0 (RCCL over xGMI) */
        const double tx = now_s();
        void *p64 = NULL, *p32 = NULL;
        uint64_t n64 = 0, n32 = 0;
        chk(itx_engine_partial_buffers(eng, &p64, &p32), "itx_engine_partial_buffers");
        chk(itx_engine_partial_size(eng, &n64, &n32), "itx_engine_partial_size");
        chk(itx_engine_sync(eng), "itx_engine_sync");
        chk(itx_engine_export_partial(eng, p64, p32, NULL), "itx_engine_export_partial");
        uint64_t meta[4] = {hc ? hc->diff_subfam : 0, hc ? hc->dup_unique : 0, boundary_missed, ends};
        itx_comm *comm = NULL;
        const double tc = now_s();
        int comm_mode = multi_comm_mode();
        int crc = ITX_OK;
        if (comm_mode == ITX_COMM_RCCL) {
            if (!early_comm.on) {                                    /* not under way since the start of the run: make it now, same thread function */
                if (pthread_create(&early_comm.th, NULL, early_comm_main, NULL) == 0) {
                    early_comm.on = 1;
                } else {
                    early_comm.rc = ITX_E_STATE;
                    snprintf(early_comm.err, sizeof early_comm.err, "no thread for the communicator");
                    comm_marker_write(0);
                    early_comm.done = 1;
                }
            }
            const char *te = getenv("ITX_COMM_TIMEOUT");
            int agreed;
            if (world > 1) {
                agreed = comm_agree(te && atof(te) > 0 ? atof(te) : 900.0);
            } else {                                                 /* ITX_COMM_SELFTEST: a job of one rank agrees with itself */
                if (early_comm.on) pthread_join(early_comm.th, NULL);
                early_comm.on = 0;
                agreed = early_comm.rc == ITX_OK;
            }
            if (agreed < 0) die("rank %d: the other ranks never said whether they have a communicator (a rank of the job has died?)", rank);
            if (agreed == 1 || __atomic_load_n(&early_comm.done, __ATOMIC_ACQUIRE)) {
                if (early_comm.on) pthread_join(early_comm.th, NULL);
                early_comm.on = 0;
                crc = early_comm.rc;
                comm = early_comm.comm;
            } else {
                early_comm.on = 0;                                   /* still inside ncclCommInitRank, waiting for a rank that will not come: left behind */
                crc = ITX_E_STATE;
            }
            if (agreed == 0) {
                /* some rank has no RCCL communicator (no usable network interface for its bootstrap, its device, its library):
                 * every rank has seen the same markers and hands its partial over through files — slower, same sums */
                if (early_comm.err[0]) warnf("[iteres] note: no RCCL communicator (%s); the ranks exchange through files instead", early_comm.err);
                else warnf("[iteres] note: another rank has no RCCL communicator; the ranks exchange through files instead");
                if (crc == ITX_OK && comm) itx_comm_destroy(comm);
                comm = NULL;
                comm_mode = ITX_COMM_FILE;
                crc = itx_comm_create(rank, world, multi_device(), multi_comm_id(), comm_mode, &comm);
            }
        } else {
            crc = itx_comm_create(rank, world, multi_device(), multi_comm_id(), comm_mode, &comm);
        }
        chk(crc, "itx_comm_create");
        const double ty = now_s();
        chk(itx_comm_reduce_sum(comm, p64, n64, p32, n32, meta, 4, NULL), "itx_comm_reduce_sum");
        if (timing && rank == 0)
            fprintf(stderr, "[itx timing] exchange (%s): export %.3f s, communicator %.3f s, reduce of %.1f MB per rank (waits for the slowest rank) %.3f s\n",
                    comm_mode == ITX_COMM_FILE ? "files" : "RCCL", tc - tx, ty - tc, (double)(n64 * 8 + n32 * 4) / 1e6, now_s() - ty);
        itx_comm_destroy(comm);
        if (rank == 0 && world > 1) {
            comm_markers_remove();                                   /* every rank has read them: its partial is here */
            /* RCCL was given up for the files: the communicator id this rank may have written for it (its thread still sits in
             * ncclCommInitRank, waiting for a rank that will not come) goes too */
            if (comm_mode == ITX_COMM_FILE && multi_comm_mode() == ITX_COMM_RCCL) unlink(multi_comm_id());
        }
        if (rank > 0) {                                              /* handed over: rank 0 writes the files */
            fflush(NULL);
            _exit(0);
        }
        if (hc) {
            hc->diff_subfam = meta[0];
            hc->dup_unique = meta[1];
        }
        boundary_missed = meta[2];
        if (shared && !boundary_missed) fprintf(stderr, "\r* Processed read ends: %llu\n", (unsigned long long)meta[3]);
        if (!boundary_missed) {
            g_reduced_u64 = p64;
            g_reduced_u32 = p32;
        }
    }
    }
    if (timing && veto_on)
        fprintf(stderr, "[itx timing] XA veto: %llu batches judged on the device, %llu by the host\n", veto_dev_batches, veto_host_batches);
    names_free(&warned);
    names_free(&chr_names);
    free(arg);
    if (bed_f) fclose(bed_f);
    if (bed_uniq_f) fclose(bed_uniq_f);
    dup_set_free(dups);
    if (dd) {
        uint64_t du = 0, dropped = 0, keys = 0;
        chk(itx_dedup_counts(dd, &du, &dropped, &keys), "itx_dedup_counts");
        if (hc) hc->dup_unique = du;
        if (timing) fprintf(stderr, "[itx timing] -R on the device: %llu records dropped (%llu of them MAPQ >= -Q), %llu keys, %.3f s in its kernels\n",
                            (unsigned long long)dropped, (unsigned long long)du, (unsigned long long)keys, t_dedup);
        itx_dedup_destroy(dd);
    }
    xa_index_free(xi);
    itx_xaveto_destroy(xv);
    free(iv);
    free(live);

    if (want_qnames && locus_names) {
        /* names per locus in BAM order (generic.c:1729 reverses the head-inserted list back to file order) */
        char **out = xcalloc(rm->n_rows ? rm->n_rows : 1, sizeof(char *));
        size_t *len = xcalloc(rm->n_rows ? rm->n_rows : 1, sizeof(size_t));
        for (size_t i = 0; i < hn.n; i++) len[hn.v[i].row] += strlen(hn.v[i].name) + 1;
        for (size_t r = 0; r < rm->n_rows; r++)
            if (len[r]) {
                out[r] = xmalloc(len[r] + 1);
                out[r][0] = 0;
                len[r] = 0;
            }
        for (size_t i = 0; i < hn.n; i++) {
            char *dst = out[hn.v[i].row] + len[hn.v[i].row];
            const size_t k = strlen(hn.v[i].name);
            if (len[hn.v[i].row]) *dst++ = ',', len[hn.v[i].row]++;
            memcpy(dst, hn.v[i].name, k + 1);
            len[hn.v[i].row] += k;
            free(hn.v[i].name);
        }
        free(len);
        *locus_names = out;
    }
    free(hn.v);
    for (int k = 0; k < 2; k++) {
        free(side[k].qname);
        free(side[k].xa);
        free(side[k].nm);
    }
    /* The decoder's device memory (17 GB for a big input) goes back NOW, not when the process ends: the driver clears released
     * memory in the background, and the next command's reservation of the same 17 GB waits for whatever is still uncleared —
     * 0.7 s now and then when one run followed another at once. With the files still to be written the driver has half a
     * second's head start, and the process's own exit has less to tear down. */
    if (g_inflater && !getenv("ITX_KEEP_DECODER")) {
        const double tr = now_s();
        itx_timing_report();
        itx_inflater_destroy(g_inflater);
        g_inflater = NULL;
        aln_use_device(NULL);
        if (timing) fprintf(stderr, "[itx timing] decoder released %.3f s\n", now_s() - tr);
    }
    *eng_out = eng;
    *tab_out = tab;
}
