// itx_comm.hip — the ONE exchange of the multi-GPU path: at end of stream every rank (one process per GPU) holds a
// compact partial in HBM (include/iteres_amd.h: itx_engine_export_partial); it is sum-reduced onto rank 0, the rank that
// writes the files. All outputs are commutative integer sums, so nothing else crosses ranks.
//
//   ITX_COMM_RCCL  ncclReduce over xGMI (RCCL), one call per buffer, in place on the root. librccl is opened with dlopen
//                  when a communicator is first asked for: single-GPU runs never load it. The unique id travels through a
//                  file: rank 0 writes <id_path>, the others wait for it.
//   ITX_COMM_FILE  the partials travel through files next to <id_path> and are added on the host by rank 0: for ranks that
//                  share a device (a rehearsal on a one-GPU box — two RCCL ranks cannot use the same GPU) and as a
//                  last resort without RCCL. Same sums.
//
// A small host vector `meta` is summed the same way (the host's own counters of the record loop, boundary verdicts).
#include "itx_common.h"

#include <dlfcn.h>
#include <errno.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

struct itx_comm {
    int rank, world, device, mode;
    std::string id_path;
    double timeout_s;
    // RCCL
    void *lib;
    ncclComm_t comm;
    ncclResult_t (*p_get_id)(ncclUniqueId *);
    ncclResult_t (*p_init_rank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*p_reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*p_destroy)(ncclComm_t);
    const char *(*p_errstr)(ncclResult_t);
    uint64_t *d_meta;
};

#define COMM_HIP(call)                                                                                    \
    do {                                                                                                  \
        hipError_t err__ = (call);                                                                        \
        if (err__ != hipSuccess) {                                                                        \
            itx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__, __LINE__); \
            return ITX_E_NO_DEVICE;                                                                       \
        }                                                                                                 \
    } while (0)
#define COMM_NCCL(c, call)                                                                                \
    do {                                                                                                  \
        ncclResult_t r__ = (call);                                                                        \
        if (r__ != ncclSuccess) {                                                                         \
            itx_set_error("%s failed: %s", #call, (c)->p_errstr ? (c)->p_errstr(r__) : "?");              \
            return ITX_E_NO_DEVICE;                                                                       \
        }                                                                                                 \
    } while (0)

static double now_s()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// waits until `path` exists (written under another name and renamed, so it is complete when it appears)
static int wait_for(const itx_comm *c, const std::string &path)
{
    const double t0 = now_s();
    struct stat sb;
    unsigned spins = 0;
    while (stat(path.c_str(), &sb) != 0) {
        if (now_s() - t0 > c->timeout_s) {
            itx_set_error("rank %d: gave up waiting for %s after %.0f s (a rank of the job has died?)", c->rank, path.c_str(), c->timeout_s);
            return ITX_E_STATE;
        }
        usleep(spins++ < 2000 ? 200 : 2000);
    }
    return ITX_OK;
}

static int write_whole(const std::string &path, const void *a, size_t na, const void *b, size_t nb, const void *m, size_t nm)
{
    const std::string tmp = path + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) {
        itx_set_error("cannot write %s: %s", tmp.c_str(), strerror(errno));
        return ITX_E_STATE;
    }
    const bool ok = (na == 0 || fwrite(a, 1, na, f) == na) && (nb == 0 || fwrite(b, 1, nb, f) == nb) && (nm == 0 || fwrite(m, 1, nm, f) == nm);
    if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) {
        itx_set_error("cannot write %s: %s", path.c_str(), strerror(errno));
        return ITX_E_STATE;
    }
    return ITX_OK;
}

// rank 0 could not get as far as a communicator id: the ranks waiting for the id file find this instead and fall back with it
// (to the exchange through files) at once, not after their timeout
static void tell_no_id(const itx_comm *c)
{
    if (c->rank == 0 && c->world > 1 && !c->id_path.empty()) (void)write_whole(c->id_path, "NOID", 4, nullptr, 0, nullptr, 0);
}

extern "C" int itx_comm_create(int rank, int world, int device, const char *id_path, int mode, itx_comm **out)
{
    if (!out || world < 1 || rank < 0 || rank >= world || (world > 1 && (!id_path || !*id_path)) || (mode != ITX_COMM_RCCL && mode != ITX_COMM_FILE)) {
        itx_set_error("itx_comm_create: bad argument");
        return ITX_E_ARG;
    }
    *out = nullptr;
    itx_comm *c = new itx_comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    c->mode = mode;
    c->id_path = id_path ? id_path : "";
    const char *te = getenv("ITX_COMM_TIMEOUT");
    c->timeout_s = te && atof(te) > 0 ? atof(te) : 900.0;
    c->lib = nullptr;
    c->comm = nullptr;
    c->d_meta = nullptr;
    // a single rank needs no communicator — unless ITX_COMM_SELFTEST asks for the whole RCCL bring-up anyway (the only way
    // to exercise it on a one-GPU box: library, entry points, id hand-over, a one-rank reduce)
    if ((world == 1 && !(mode == ITX_COMM_RCCL && getenv("ITX_COMM_SELFTEST") && id_path && *id_path)) || mode == ITX_COMM_FILE) {
        *out = c;
        return ITX_OK;
    }
    if (!getenv("ITX_COMM_NO_RCCL")) {                     // (tests: a rank that cannot load the library)
        c->lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!c->lib) c->lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    }
    if (!c->lib) {
        const char *why = dlerror();
        itx_set_error("cannot load librccl.so.1: %s", why ? why : "switched off (ITX_COMM_NO_RCCL)");
        tell_no_id(c);
        delete c;
        return ITX_E_NO_DEVICE;
    }
    c->p_get_id = (decltype(c->p_get_id))dlsym(c->lib, "ncclGetUniqueId");
    c->p_init_rank = (decltype(c->p_init_rank))dlsym(c->lib, "ncclCommInitRank");
    c->p_reduce = (decltype(c->p_reduce))dlsym(c->lib, "ncclReduce");
    c->p_destroy = (decltype(c->p_destroy))dlsym(c->lib, "ncclCommDestroy");
    c->p_errstr = (decltype(c->p_errstr))dlsym(c->lib, "ncclGetErrorString");
    if (!c->p_get_id || !c->p_init_rank || !c->p_reduce || !c->p_destroy) {
        itx_set_error("librccl lacks an entry point this path needs");
        tell_no_id(c);
        delete c;
        return ITX_E_NO_DEVICE;
    }
    hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) {
        itx_set_error("hipSetDevice(%d) failed: %s", device, hipGetErrorString(he));
        tell_no_id(c);
        delete c;
        return ITX_E_NO_DEVICE;
    }
    ncclUniqueId id;
    memset(&id, 0, sizeof id);
    if (rank == 0) {
        ncclResult_t r = c->p_get_id(&id);
        if (r != ncclSuccess) {
            itx_set_error("ncclGetUniqueId failed: %s", c->p_errstr ? c->p_errstr(r) : "?");
            tell_no_id(c);
            delete c;
            return ITX_E_NO_DEVICE;
        }
        int rc = write_whole(c->id_path, &id, sizeof id, nullptr, 0, nullptr, 0);
        if (rc != ITX_OK) {
            delete c;
            return rc;
        }
    } else {
        int rc = wait_for(c, c->id_path);
        if (rc != ITX_OK) {
            delete c;
            return rc;
        }
        FILE *f = fopen(c->id_path.c_str(), "rb");
        const size_t got = f ? fread(&id, 1, sizeof id, f) : 0;
        const bool ok = got == sizeof id;
        if (f) fclose(f);
        if (!ok) {
            if (got == 4 && memcmp(&id, "NOID", 4) == 0) itx_set_error("rank 0 could not make a communicator id");
            else itx_set_error("cannot read the communicator id from %s", c->id_path.c_str());
            delete c;
            return ITX_E_STATE;
        }
    }
    ncclResult_t r = c->p_init_rank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        itx_set_error("ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device, c->p_errstr ? c->p_errstr(r) : "?");
        delete c;
        return ITX_E_NO_DEVICE;
    }
    *out = c;
    return ITX_OK;
}

extern "C" void itx_comm_destroy(itx_comm *c)
{
    if (!c) return;
    if (c->d_meta) (void)hipFree(c->d_meta);
    if (c->comm && c->p_destroy) (void)c->p_destroy(c->comm);
    if (c->rank == 0 && c->comm && c->mode == ITX_COMM_RCCL) (void)unlink(c->id_path.c_str());
    // the library stays loaded: unloading RCCL while its proxy threads wind down is not worth the risk
    delete c;
}

/* Sum over ranks of the two device buffers (n64 x uint64, n32 x uint32; wrap-around sums like the reference's unsigned
 * counters) and of the host vector meta[n_meta], left on rank 0 (in place; the other ranks' buffers are undefined
 * afterwards). Enqueued behind `stream`'s work; returns when the result is there. */
extern "C" int itx_comm_reduce_sum(itx_comm *c, void *d_u64, size_t n64, void *d_u32, size_t n32, uint64_t *meta, size_t n_meta, void *stream)
{
    if (!c || (n64 && !d_u64) || (n32 && !d_u32) || (n_meta && !meta)) {
        itx_set_error("itx_comm_reduce_sum: bad argument");
        return ITX_E_ARG;
    }
    COMM_HIP(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    if (c->world == 1 && !c->comm) {
        COMM_HIP(hipStreamSynchronize(st));
        return ITX_OK;
    }
    if (c->mode == ITX_COMM_RCCL) {
        if (n_meta) {
            if (!c->d_meta) COMM_HIP(hipMalloc((void **)&c->d_meta, sizeof(uint64_t) * 64));
            if (n_meta > 64) return ITX_E_ARG;
            COMM_HIP(hipMemcpyAsync(c->d_meta, meta, sizeof(uint64_t) * n_meta, hipMemcpyHostToDevice, st));
        }
        if (n64) COMM_NCCL(c, c->p_reduce(d_u64, d_u64, n64, ncclUint64, ncclSum, 0, c->comm, st));
        if (n32) COMM_NCCL(c, c->p_reduce(d_u32, d_u32, n32, ncclUint32, ncclSum, 0, c->comm, st));
        if (n_meta) COMM_NCCL(c, c->p_reduce(c->d_meta, c->d_meta, n_meta, ncclUint64, ncclSum, 0, c->comm, st));
        if (n_meta && c->rank == 0) COMM_HIP(hipMemcpyAsync(meta, c->d_meta, sizeof(uint64_t) * n_meta, hipMemcpyDeviceToHost, st));
        COMM_HIP(hipStreamSynchronize(st));
        return ITX_OK;
    }
    // ---- through files
    COMM_HIP(hipStreamSynchronize(st));
    std::vector<uint64_t> h64(n64 + 1);
    std::vector<uint32_t> h32(n32 + 1);
    if (n64) COMM_HIP(hipMemcpy(h64.data(), d_u64, sizeof(uint64_t) * n64, hipMemcpyDeviceToHost));
    if (n32) COMM_HIP(hipMemcpy(h32.data(), d_u32, sizeof(uint32_t) * n32, hipMemcpyDeviceToHost));
    if (c->rank != 0) {
        int rc = write_whole(c->id_path + ".part" + std::to_string(c->rank), h64.data(), sizeof(uint64_t) * n64, h32.data(), sizeof(uint32_t) * n32, meta,
                             sizeof(uint64_t) * n_meta);
        if (rc != ITX_OK) return rc;
        // the root takes the file away when it has read it: nothing of this job is left behind
        return ITX_OK;
    }
    std::vector<uint64_t> p64(n64 + 1), pm(n_meta + 1);
    std::vector<uint32_t> p32(n32 + 1);
    for (int r = 1; r < c->world; r++) {
        const std::string path = c->id_path + ".part" + std::to_string(r);
        int rc = wait_for(c, path);
        if (rc != ITX_OK) return rc;
        FILE *f = fopen(path.c_str(), "rb");
        const bool ok = f && (n64 == 0 || fread(p64.data(), sizeof(uint64_t), n64, f) == n64) && (n32 == 0 || fread(p32.data(), sizeof(uint32_t), n32, f) == n32) &&
                        (n_meta == 0 || fread(pm.data(), sizeof(uint64_t), n_meta, f) == n_meta) && fgetc(f) == EOF;
        if (f) fclose(f);
        (void)unlink(path.c_str());
        if (!ok) {
            itx_set_error("the partial of rank %d (%s) does not have the size this rank expects", r, path.c_str());
            return ITX_E_STATE;
        }
        for (size_t i = 0; i < n64; i++) h64[i] += p64[i];
        for (size_t i = 0; i < n32; i++) h32[i] += p32[i];
        for (size_t i = 0; i < n_meta; i++) meta[i] += pm[i];
    }
    if (n64) COMM_HIP(hipMemcpy(d_u64, h64.data(), sizeof(uint64_t) * n64, hipMemcpyHostToDevice));
    if (n32) COMM_HIP(hipMemcpy(d_u32, h32.data(), sizeof(uint32_t) * n32, hipMemcpyHostToDevice));
    return ITX_OK;
}
