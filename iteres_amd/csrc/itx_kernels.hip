// itx_kernels.hip — gfx950 kernels of the iteres hot path.
//
//   k_classify_atomic   one record per lane: derive -> classify -> (optionally) accumulate with global
//                       atomics. This is the ITX_ACCUM_ATOMIC path and the classify-only entry point.
//   k_finalize_rep      per repName: read count = sum of range starts, coverage = prefix sum of
//                       (starts - ends) over the name's slots.
//
// All arithmetic is integer except the one f32 ratio of generic.c:296-301 (__fdiv_rn, IEEE).
#include "itx_device.h"

#define ITX_BLOCK 256

__device__ __forceinline__ void wave_count_to_lds(uint32_t *s_cnt, uint32_t cntbits)
{
    // 11 live counters (generic.c:1048-1060; 8 and 12 never set here)
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        if (k == 8) continue;
        const unsigned long long m = __ballot((cntbits >> k) & 1u);
        if (lane == 0 && m) atomicAdd(&s_cnt[k], (uint32_t)__popcll(m));
    }
}

template <bool ACCUM>
__global__ __launch_bounds__(ITX_BLOCK) void k_classify_atomic(ItxDevTable T, ItxRunParams P, ItxDevBatch B, size_t n,
                                                               int32_t *__restrict__ d_hit_row, uint64_t *__restrict__ u64,
                                                               uint32_t *__restrict__ u32, ItxAccumLayout L, int fc_in_lds)
{
    __shared__ uint32_t s_cnt[16];
    extern __shared__ uint32_t s_fc[];   // fam[2F] | cla[2C] when fc_in_lds
    const uint32_t nfc = 2 * (T.n_fam + T.n_cla);
    if (threadIdx.x < 16) s_cnt[threadIdx.x] = 0;
    if (ACCUM && fc_in_lds)
        for (uint32_t k = threadIdx.x; k < nfc; k += ITX_BLOCK) s_fc[k] = 0;
    __syncthreads();

    const size_t stride = (size_t)gridDim.x * ITX_BLOCK;
    const size_t n_round = (n + ITX_BLOCK - 1) / ITX_BLOCK * ITX_BLOCK;
    for (size_t i = (size_t)blockIdx.x * ITX_BLOCK + threadIdx.x; i < n_round; i += stride) {
        uint32_t cntbits = 0;
        int32_t hit = -1;
        ItxDerived d;
        d.chrom = -1;
        d.uniq = false;
        if (i < n) {
            d = itx_derive(P, T, B, i);
            cntbits = d.cntbits;
            if (d.chrom >= 0) hit = itx_classify(T, d.chrom, d.start, d.end, P.min_cov);
            if (hit >= 0) cntbits |= (1u << 9) | (d.uniq ? (1u << 10) : 0u);   // generic.c:1030-1032
            if (d_hit_row) d_hit_row[i] = hit >= 0 ? T.orig[hit] : -1;
        }
        wave_count_to_lds(s_cnt, cntbits);
        if (ACCUM && hit >= 0) {
            const ItxIv r = T.iv[hit];
            if (P.mode == ITX_MODE_STAT) {
                uint32_t first;
                const uint32_t nc = itx_cov_range(r, d.start, d.end, &first);
                if (nc) {
                    atomicAdd(&u32[L.a_all + first], 1u);
                    atomicAdd(&u32[L.b_all + first + nc], 1u);
                    if (d.uniq) {
                        atomicAdd(&u32[L.a_uniq + first], 1u);
                        atomicAdd(&u32[L.b_uniq + first + nc], 1u);
                    }
                } else {
                    atomicAdd(&u32[L.a_all + r.zslot], 1u);
                    if (d.uniq) atomicAdd(&u32[L.a_uniq + r.zslot], 1u);
                }
                const uint32_t fam = r.famcla >> 16, cla = r.famcla & 0xffffu;
                if (fc_in_lds) {
                    atomicAdd(&s_fc[fam], 1u);
                    atomicAdd(&s_fc[2 * T.n_fam + cla], 1u);
                    if (d.uniq) {
                        atomicAdd(&s_fc[T.n_fam + fam], 1u);
                        atomicAdd(&s_fc[2 * T.n_fam + T.n_cla + cla], 1u);
                    }
                } else {
                    atomicAdd((unsigned long long *)&u64[L.fam + fam], 1ull);
                    atomicAdd((unsigned long long *)&u64[L.cla + cla], 1ull);
                    if (d.uniq) {
                        atomicAdd((unsigned long long *)&u64[L.fam + T.n_fam + fam], 1ull);
                        atomicAdd((unsigned long long *)&u64[L.cla + T.n_cla + cla], 1ull);
                    }
                }
            } else {
                atomicAdd(&u32[L.locus + (uint32_t)hit], 1u);   // per sorted row; permuted to caller order in finish
            }
        }
    }
    __syncthreads();
    if (ACCUM && threadIdx.x < 16 && s_cnt[threadIdx.x])   // classify-only launches leave every accumulator alone
        atomicAdd((unsigned long long *)&u64[L.cnt + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    if (ACCUM && fc_in_lds && P.mode == ITX_MODE_STAT) {
        for (uint32_t k = threadIdx.x; k < nfc; k += ITX_BLOCK) {
            const uint32_t v = s_fc[k];
            if (v) atomicAdd((unsigned long long *)&u64[L.fam + k], (unsigned long long)v);   // fam|cla are contiguous
        }
    }
}

int itx_launch_atomic(const ItxDevTable &T, const ItxRunParams &P, const ItxDevBatch &B, size_t n, int do_accum,
                      int32_t *d_hit_row, uint64_t *u64, uint32_t *u32, const ItxAccumLayout &L, hipStream_t st)
{
    if (n == 0) return ITX_OK;
    size_t blocks = (n + ITX_BLOCK - 1) / ITX_BLOCK;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const size_t fc_bytes = (size_t)2 * (T.n_fam + T.n_cla) * sizeof(uint32_t);
    const int fc_in_lds = fc_bytes <= 48 * 1024;
    const size_t shmem = (do_accum && fc_in_lds) ? fc_bytes : 0;
    if (do_accum)
        hipLaunchKernelGGL(k_classify_atomic<true>, dim3((unsigned)blocks), dim3(ITX_BLOCK), shmem, st, T, P, B, n, d_hit_row, u64,
                           u32, L, fc_in_lds);
    else
        hipLaunchKernelGGL(k_classify_atomic<false>, dim3((unsigned)blocks), dim3(ITX_BLOCK), 0, st, T, P, B, n, d_hit_row, u64, u32,
                           L, 0);
    ITX_HIP(hipGetLastError());
    return ITX_OK;
}

// ---------------------------------------------------------------------------------------------------
// finalize: one workgroup per repName.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

__global__ __launch_bounds__(ITX_BLOCK) void k_finalize_rep(const uint32_t *__restrict__ covslot, const uint64_t *__restrict__ covoff,
                                                            const uint32_t *__restrict__ u32, ItxAccumLayout L, uint32_t n_rep,
                                                            uint64_t *__restrict__ rep_out, uint32_t *__restrict__ cov,
                                                            uint32_t *__restrict__ cov_uniq)
{
    __shared__ uint32_t s_w[2][ITX_BLOCK / 64];
    __shared__ unsigned long long s_sum[2];
    const uint32_t r = blockIdx.x;
    if (r >= n_rep) return;
    const uint32_t s0 = covslot[r], s1 = covslot[r + 1];   // len + 1 slots
    const uint32_t len = s1 - s0 - 1;
    const uint64_t o = covoff[r];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x < 2) s_sum[threadIdx.x] = 0;
    __syncthreads();
    uint32_t carry_a = 0, carry_u = 0;
    unsigned long long sum_a = 0, sum_u = 0;
    for (uint32_t base = 0; base < len + 1; base += ITX_BLOCK) {
        const uint32_t j = base + threadIdx.x;
        uint32_t aa = 0, au = 0, da = 0, du = 0;
        if (j <= len) {
            aa = u32[L.a_all + s0 + j];
            au = u32[L.a_uniq + s0 + j];
            da = aa - u32[L.b_all + s0 + j];
            du = au - u32[L.b_uniq + s0 + j];
        }
        sum_a += aa;
        sum_u += au;
        uint32_t pa = wave_incl_scan(da), pu = wave_incl_scan(du);
        if (lane == 63) {
            s_w[0][w] = pa;
            s_w[1][w] = pu;
        }
        __syncthreads();
        uint32_t offa = carry_a, offu = carry_u;
        for (int k = 0; k < w; k++) {
            offa += s_w[0][k];
            offu += s_w[1][k];
        }
        pa += offa;
        pu += offu;
        if (j < len) {
            if (cov) cov[o + j] = pa;
            if (cov_uniq) cov_uniq[o + j] = pu;
        }
        for (int k = 0; k < ITX_BLOCK / 64; k++) {
            carry_a += s_w[0][k];
            carry_u += s_w[1][k];
        }
        __syncthreads();
    }
    // read counts = number of range starts recorded anywhere in the name's slots
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
        sum_a += __shfl_down(sum_a, o2, 64);
        sum_u += __shfl_down(sum_u, o2, 64);
    }
    if (lane == 0) {
        atomicAdd(&s_sum[0], sum_a);
        atomicAdd(&s_sum[1], sum_u);
    }
    __syncthreads();
    if (threadIdx.x == 0 && rep_out) {
        rep_out[r] = s_sum[0];
        rep_out[n_rep + r] = s_sum[1];
    }
}

int itx_launch_finalize(const itx_table *t, const uint64_t *u64, const uint32_t *u32, const ItxAccumLayout &L,
                        uint64_t *d_rep_out, uint32_t *d_cov, uint32_t *d_cov_uniq, hipStream_t st)
{
    (void)u64;
    if (t->n_rep == 0) return ITX_OK;
    hipLaunchKernelGGL(k_finalize_rep, dim3(t->n_rep), dim3(ITX_BLOCK), 0, st, t->d_covslot, t->d_covoff, u32, L, t->n_rep, d_rep_out,
                       d_cov, d_cov_uniq);
    ITX_HIP(hipGetLastError());
    return ITX_OK;
}
