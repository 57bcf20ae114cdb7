// itx_inflate.hip — BGZF blocks inflated on the device: one wavefront per block (itx_inflate_core.h), thousands of blocks
// per launch. Replaces the reference's sequential zlib inflate of every 64 KiB block (cussamtools/bgzf.c:367-397
// inflate_block, reached from bgzf_read -> bgzf_read_block, bgzf.c:425-521): only compressed bytes cross PCIe on the way in.
//
// HBM layout: the caller's compressed chunk is mirrored at the same offsets in d_comp (so a block's position keeps its
// alignment), the inflated bytes of all blocks are contiguous in d_out at the offsets the caller computed from the ISIZE
// trailers. A call is cut into groups of blocks that alternate between two streams: copy-in, kernel and copy-out of one
// group overlap the neighbours'.
#include "itx_common.h"

#define ITXI_WAVE 64u
#define ITXI_FN static __device__ inline
#define ITXI_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(x)))
#define ITXI_LOADW(w, i) ((w)[(i)])
#define ITXI_LOADB(p, i) ((p)[(i)])
// a far match reads bytes this wave stored earlier through other lanes: make the stores visible first
#define ITXI_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent")
#include "itx_inflate_core.h"

#include <stdlib.h>
#include <string.h>

#define BGZF_HEADER 18u      /* gzip header with the one "BC" extra field (bgzf.c:401-411) */
#define BGZF_TRAILER 8u      /* CRC32 + ISIZE */

__global__ __launch_bounds__(64) void k_inflate(const uint32_t *__restrict__ comp, const itx_bgzf_block *__restrict__ blk, uint32_t first, uint32_t n,
                                                uint8_t *__restrict__ out, uint8_t *__restrict__ status)
{
    __shared__ ItxiLds S;
    const uint32_t b = first + blockIdx.x;
    if (blockIdx.x >= n) return;
    const uint32_t coff = blk[b].coff, csize = blk[b].csize, uoff = blk[b].uoff, usize = blk[b].usize;
    int rc = ITXI_E_INPUT;
    if (csize >= BGZF_HEADER + BGZF_TRAILER + 2u) rc = itxi_block(S, comp, coff + BGZF_HEADER, coff + csize - BGZF_TRAILER, out, uoff, usize, threadIdx.x);
    if (threadIdx.x == 0) status[b] = (uint8_t)rc;
}

struct itx_inflater {
    int device;
    hipStream_t st[2];
    uint8_t *d_comp, *d_out, *d_status;
    itx_bgzf_block *d_blk;
    size_t comp_cap, out_cap, status_cap, blk_cap;
};

#define INF_HIP(call)                                                                                     \
    do {                                                                                                  \
        hipError_t err__ = (call);                                                                        \
        if (err__ != hipSuccess) {                                                                        \
            itx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__, __LINE__); \
            return ITX_E_NO_DEVICE;                                                                       \
        }                                                                                                 \
    } while (0)

extern "C" int itx_inflater_create(int device, itx_inflater **out)
{
    if (!out) return ITX_E_ARG;
    *out = nullptr;
    INF_HIP(hipSetDevice(device));
    itx_inflater *h = (itx_inflater *)calloc(1, sizeof *h);
    if (!h) return ITX_E_NOMEM;
    h->device = device;
    for (int k = 0; k < 2; k++) INF_HIP(hipStreamCreateWithFlags(&h->st[k], hipStreamNonBlocking));
    *out = h;
    return ITX_OK;
}

extern "C" void itx_inflater_destroy(itx_inflater *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    for (int k = 0; k < 2; k++)
        if (h->st[k]) {
            (void)hipStreamSynchronize(h->st[k]);
            (void)hipStreamDestroy(h->st[k]);
        }
    (void)hipFree(h->d_comp);
    (void)hipFree(h->d_out);
    (void)hipFree(h->d_status);
    (void)hipFree(h->d_blk);
    free(h);
}

extern "C" void *itx_pinned_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

extern "C" void itx_pinned_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

template <typename T> static int grow(T **p, size_t *cap, size_t need)
{
    if (need <= *cap) return ITX_OK;
    if (*p) INF_HIP(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    const size_t want = need + need / 4;
    INF_HIP(hipMalloc((void **)p, want * sizeof(T)));
    *cap = want;
    return ITX_OK;
}

extern "C" int itx_inflate_bgzf(itx_inflater *h, const void *comp, size_t comp_len, const itx_bgzf_block *blk, size_t n_blk, void *out, size_t out_len,
                                uint8_t *status)
{
    if (!h || !comp || !blk || !out || !status) return ITX_E_ARG;
    if (n_blk == 0) return ITX_OK;
    if (comp_len > 0xfffffff0u || out_len > 0xfffffff0u || n_blk > 0x7fffffffu) return ITX_E_LIMIT;
    // what the kernel assumes about every block, checked here: inside the buffers, outputs disjoint and in order
    size_t uat = 0;
    for (size_t i = 0; i < n_blk; i++) {
        const itx_bgzf_block &b = blk[i];
        if ((size_t)b.coff + b.csize > comp_len || b.csize < BGZF_HEADER + BGZF_TRAILER || b.uoff != uat || (size_t)b.uoff + b.usize > out_len ||
            b.usize > 65536u) {
            itx_set_error("itx_inflate_bgzf: block %zu does not fit its buffers", i);
            return ITX_E_ARG;
        }
        uat += b.usize;
    }
    INF_HIP(hipSetDevice(h->device));
    int rc;
    if ((rc = grow(&h->d_comp, &h->comp_cap, comp_len + 64)) != ITX_OK) return rc;
    if ((rc = grow(&h->d_out, &h->out_cap, out_len + 64)) != ITX_OK) return rc;
    if ((rc = grow(&h->d_status, &h->status_cap, n_blk)) != ITX_OK) return rc;
    if ((rc = grow(&h->d_blk, &h->blk_cap, n_blk)) != ITX_OK) return rc;
    INF_HIP(hipMemcpyAsync(h->d_blk, blk, n_blk * sizeof *blk, hipMemcpyHostToDevice, h->st[0]));
    INF_HIP(hipStreamSynchronize(h->st[0]));
    // groups of blocks, alternating streams
    const size_t per = n_blk < 4096 ? (n_blk + 1) / 2 : 2048;
    int s = 0;
    for (size_t b0 = 0; b0 < n_blk; b0 += per, s ^= 1) {
        const size_t b1 = b0 + per < n_blk ? b0 + per : n_blk;
        const size_t c0 = blk[b0].coff & ~(size_t)3, c1 = (size_t)blk[b1 - 1].coff + blk[b1 - 1].csize;
        INF_HIP(hipMemcpyAsync(h->d_comp + c0, (const uint8_t *)comp + c0, c1 - c0, hipMemcpyHostToDevice, h->st[s]));
        hipLaunchKernelGGL(k_inflate, dim3((unsigned)(b1 - b0)), dim3(64), 0, h->st[s], (const uint32_t *)h->d_comp, h->d_blk, (uint32_t)b0,
                           (uint32_t)(b1 - b0), h->d_out, h->d_status);
        INF_HIP(hipGetLastError());
        const size_t u0 = blk[b0].uoff, u1 = (size_t)blk[b1 - 1].uoff + blk[b1 - 1].usize;
        if (u1 > u0) INF_HIP(hipMemcpyAsync((uint8_t *)out + u0, h->d_out + u0, u1 - u0, hipMemcpyDeviceToHost, h->st[s]));
        INF_HIP(hipMemcpyAsync(status + b0, h->d_status + b0, b1 - b0, hipMemcpyDeviceToHost, h->st[s]));
    }
    INF_HIP(hipStreamSynchronize(h->st[0]));
    INF_HIP(hipStreamSynchronize(h->st[1]));
    return ITX_OK;
}
