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
#define ITXI_BCAST(v, j) ((uint32_t)__builtin_amdgcn_readlane((int32_t)(v), (int32_t)(j)))
#define ITXI_AT(p, i) (p)[(i) * 64u + ln]          /* a decoder's table element i: lane-interleaved (bank = lane) */
#define ITXI_LOADW(w, i) ((w)[(i)])
#define ITXI_LOADB(p, i) ((p)[(i)])
// A far match reads bytes this wave stored earlier through other lanes. Workgroup scope is all it takes — the wave's
// stores have to be acknowledged before its loads go out (s_waitcnt vmcnt(0)); the lines read are whole stripes written
// once, so no cached copy can be stale. (An agent-scope fence here writes the L2 back: measured 4 500 cycles per token.)
#define ITXI_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")
#include "itx_inflate_core.h"

#include <stdlib.h>
#include <string.h>

#define BGZF_HEADER 18u      /* gzip header with the one "BC" extra field (bgzf.c:401-411) */
#define BGZF_TRAILER 8u      /* CRC32 + ISIZE */
#define LIT_STRIDE (ITXI_MAX_BLOCK + 16u)         /* bytes of literal scratch per block (the stage reads 16 at a time) */
#define TOK_STRIDE (2u * ITXI_MAX_TOK)            /* words of token scratch per block */

// pass 1: lane = block. meta[3b] = status, [3b+1] = literals, [3b+2] = matches
__global__ __launch_bounds__(64) void k_tokens(const uint32_t *__restrict__ comp, const itx_bgzf_block *__restrict__ blk, uint32_t n, uint8_t *__restrict__ lit,
                                               uint32_t *__restrict__ tok, uint32_t *__restrict__ meta)
{
    __shared__ uint16_t s_lsym[288 * 64], s_dsym[32 * 64], s_offs[16 * 64];
    __shared__ uint8_t s_lens[352 * 64];
    const uint32_t ln = threadIdx.x, b = blockIdx.x * 64u + ln;
    if (b >= n) return;
    const uint32_t coff = blk[b].coff, csize = blk[b].csize, usize = blk[b].usize;
    ItxiTab T{s_lsym, s_dsym, s_offs, s_lens};
    ItxiTokens K{lit + (size_t)b * LIT_STRIDE, tok + (size_t)b * TOK_STRIDE, 0, 0};
    int rc = ITXI_E_INPUT;
    if (csize >= BGZF_HEADER + BGZF_TRAILER + 2u && usize <= ITXI_MAX_BLOCK)
        rc = itxi_tokens(T, ln, comp, coff + BGZF_HEADER, coff + csize - BGZF_TRAILER, usize, K);
    meta[3 * b] = (uint32_t)rc;
    meta[3 * b + 1] = K.n_lit;
    meta[3 * b + 2] = K.n_tok;
}

// pass 2: wave = block
__global__ __launch_bounds__(64) void k_resolve(const itx_bgzf_block *__restrict__ blk, uint32_t first, uint32_t n, const uint8_t *__restrict__ lit,
                                                const uint32_t *__restrict__ tok, const uint32_t *__restrict__ meta, uint8_t *__restrict__ out,
                                                uint8_t *__restrict__ status)
{
    __shared__ uint32_t s_ring[ITXI_RING / 4], s_stage[ITXI_LSTAGE / 4];
    if (blockIdx.x >= n) return;
    const uint32_t b = first + blockIdx.x;
    int rc = (int)meta[3 * b];
    if (rc == ITXI_OK)
        rc = itxi_resolve(s_ring, s_stage, lit + (size_t)b * LIT_STRIDE, tok + (size_t)b * TOK_STRIDE, meta[3 * b + 1], meta[3 * b + 2], out, blk[b].uoff, blk[b].usize,
                          threadIdx.x);
    if (threadIdx.x == 0) status[b] = (uint8_t)rc;
}

struct itx_inflater {
    int device;
    hipStream_t st[2];
    uint8_t *d_comp, *d_out, *d_status, *d_lit;
    uint32_t *d_tok, *d_meta;
    itx_bgzf_block *d_blk;
    size_t comp_cap, out_cap, status_cap, blk_cap, lit_cap, tok_cap, meta_cap;
    hipEvent_t ev[4];
    float ms_tokens, ms_resolve;
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
    for (int k = 0; k < 4; k++) INF_HIP(hipEventCreate(&h->ev[k]));
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
    (void)hipFree(h->d_lit);
    (void)hipFree(h->d_tok);
    (void)hipFree(h->d_meta);
    for (int k = 0; k < 4; k++)
        if (h->ev[k]) (void)hipEventDestroy(h->ev[k]);
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
    if ((rc = grow(&h->d_lit, &h->lit_cap, n_blk * (size_t)LIT_STRIDE)) != ITX_OK) return rc;
    if ((rc = grow(&h->d_tok, &h->tok_cap, n_blk * (size_t)TOK_STRIDE)) != ITX_OK) return rc;
    if ((rc = grow(&h->d_meta, &h->meta_cap, 3 * n_blk)) != ITX_OK) return rc;
    // pass 1 over all blocks at once (a lane per block: it takes many blocks to fill the chip)
    INF_HIP(hipMemcpyAsync(h->d_blk, blk, n_blk * sizeof *blk, hipMemcpyHostToDevice, h->st[0]));
    INF_HIP(hipMemcpyAsync(h->d_comp, comp, comp_len, hipMemcpyHostToDevice, h->st[0]));
    INF_HIP(hipEventRecord(h->ev[0], h->st[0]));
    hipLaunchKernelGGL(k_tokens, dim3((unsigned)((n_blk + 63) / 64)), dim3(64), 0, h->st[0], (const uint32_t *)h->d_comp, h->d_blk, (uint32_t)n_blk, h->d_lit,
                       h->d_tok, h->d_meta);
    INF_HIP(hipGetLastError());
    INF_HIP(hipEventRecord(h->ev[1], h->st[0]));
    INF_HIP(hipStreamSynchronize(h->st[0]));
    // pass 2 in groups that alternate between two streams: one group's copy-out overlaps the next one's kernel
    const size_t per = n_blk < 2048 ? n_blk : (n_blk + 3) / 4;
    int s = 0;
    INF_HIP(hipEventRecord(h->ev[2], h->st[0]));
    for (size_t b0 = 0; b0 < n_blk; b0 += per, s ^= 1) {
        const size_t b1 = b0 + per < n_blk ? b0 + per : n_blk;
        hipLaunchKernelGGL(k_resolve, dim3((unsigned)(b1 - b0)), dim3(64), 0, h->st[s], h->d_blk, (uint32_t)b0, (uint32_t)(b1 - b0), h->d_lit, h->d_tok, h->d_meta,
                           h->d_out, h->d_status);
        INF_HIP(hipGetLastError());
        if (b0 == 0) INF_HIP(hipEventRecord(h->ev[3], h->st[0]));
        const size_t u0 = blk[b0].uoff, u1 = (size_t)blk[b1 - 1].uoff + blk[b1 - 1].usize;
        if (u1 > u0) INF_HIP(hipMemcpyAsync((uint8_t *)out + u0, h->d_out + u0, u1 - u0, hipMemcpyDeviceToHost, h->st[s]));
        INF_HIP(hipMemcpyAsync(status + b0, h->d_status + b0, b1 - b0, hipMemcpyDeviceToHost, h->st[s]));
    }
    INF_HIP(hipStreamSynchronize(h->st[0]));
    INF_HIP(hipStreamSynchronize(h->st[1]));
    (void)hipEventElapsedTime(&h->ms_tokens, h->ev[0], h->ev[1]);
    (void)hipEventElapsedTime(&h->ms_resolve, h->ev[2], h->ev[3]);
    return ITX_OK;
}

/* device time of the two passes of the last itx_inflate_bgzf call (pass 2: its first group), milliseconds */
extern "C" int itx_inflater_last_ms(const itx_inflater *h, float *tokens_ms, float *resolve_ms)
{
    if (!h) return ITX_E_ARG;
    if (tokens_ms) *tokens_ms = h->ms_tokens;
    if (resolve_ms) *resolve_ms = h->ms_resolve;
    return ITX_OK;
}
