// itx_finalize.hip — end-of-stream kernels: raw accumulators -> compact partial (what ranks all-reduce)
// -> the arrays the reference's writers consume (generic.c:72-113, 1709-1746).
//
//   export          the accumulators already are the partial (unit read counts, D = starts - ends): device copies
//   k_finish_unit   per unit: coverage = prefix sum of D (mod 2^32, like bp_total's unsigned int),
//                   counts added to the unit's repName / repFamily / repClass
//   k_permute_locus filter mode: per-locus counts from sorted-row order to the caller's row order
#include "itx_common.h"

#define FB 256

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

__global__ __launch_bounds__(FB) void k_finish_unit(const uint32_t *__restrict__ unit_slot, const uint4 *__restrict__ unit_ids,
                                                    const uint64_t *__restrict__ unit_covoff, const uint64_t *__restrict__ p64,
                                                    const uint32_t *__restrict__ p32, uint32_t n_units, uint32_t n_slots, uint32_t n_rep,
                                                    uint32_t n_fam, uint32_t n_cla, uint64_t *__restrict__ counts,
                                                    uint32_t *__restrict__ cov, uint32_t *__restrict__ cov_uniq)
{
    __shared__ uint32_t s_w[2][FB / 64];
    const uint32_t u = blockIdx.x;
    if (u >= n_units) return;
    const uint4 ids = unit_ids[u];
    if (threadIdx.x == 0) {
        const unsigned long long ca = p64[16 + u], cu = p64[16 + (size_t)n_units + u];
        unsigned long long *c = (unsigned long long *)counts;
        if (ca) {
            atomicAdd(&c[ids.x], ca);
            atomicAdd(&c[2 * (size_t)n_rep + ids.y], ca);
            atomicAdd(&c[2 * (size_t)n_rep + 2 * (size_t)n_fam + ids.z], ca);
        }
        if (cu) {
            atomicAdd(&c[(size_t)n_rep + ids.x], cu);
            atomicAdd(&c[2 * (size_t)n_rep + n_fam + ids.y], cu);
            atomicAdd(&c[2 * (size_t)n_rep + 2 * (size_t)n_fam + n_cla + ids.z], cu);
        }
    }
    const uint32_t s0 = unit_slot[u], s1 = unit_slot[u + 1];
    const uint32_t len = s1 - s0 - 1;                       // the extra slot is not part of the coverage
    const uint64_t o = unit_covoff[u];
    const bool solo = ids.w != 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t carry_a = 0, carry_u = 0;
    for (uint32_t base = 0; base < len; base += FB) {
        const uint32_t j = base + threadIdx.x;
        uint32_t da = 0, du = 0;
        if (j < len) {
            da = p32[s0 + j];
            du = p32[(size_t)n_slots + s0 + j];
        }
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
            if (solo) {
                cov[o + j] = pa;
                cov_uniq[o + j] = pu;
            } else {                                        // a name split over several units: sum them
                if (pa) atomicAdd(&cov[o + j], pa);
                if (pu) atomicAdd(&cov_uniq[o + j], pu);
            }
        }
        for (int k = 0; k < FB / 64; k++) {
            carry_a += s_w[0][k];
            carry_u += s_w[1][k];
        }
        __syncthreads();
    }
}

__global__ void k_permute_locus(const uint32_t *__restrict__ in, const int32_t *__restrict__ orig, uint32_t n,
                                uint32_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[orig[i]] = in[i];
}

int itx_launch_export(const itx_table *t, int mode, const uint64_t *u64, const uint32_t *u32, const ItxAccumLayout &L, uint64_t *p64,
                      uint32_t *p32, hipStream_t st)
{
    if (mode == ITX_MODE_STAT) {
        // the accumulators are kept in the partial's own form: the counters as they are, the two D arrays without their padding
        ITX_HIP(hipMemcpyAsync(p64, u64, (16 + 2 * (size_t)t->n_units) * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
        if (t->n_slots) {
            ITX_HIP(hipMemcpyAsync(p32, u32 + L.d_all, (size_t)t->n_slots * 4, hipMemcpyDeviceToDevice, st));
            ITX_HIP(hipMemcpyAsync(p32 + t->n_slots, u32 + L.d_uniq, (size_t)t->n_slots * 4, hipMemcpyDeviceToDevice, st));
        }
    } else {
        ITX_HIP(hipMemsetAsync(p64, 0, (16 + 2 * (size_t)t->n_units) * sizeof(uint64_t), st));
        ITX_HIP(hipMemcpyAsync(p64, u64, 16 * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
        if (t->n_rows) ITX_HIP(hipMemcpyAsync(p32, u32 + L.locus, (size_t)t->n_rows * 4, hipMemcpyDeviceToDevice, st));
    }
    return ITX_OK;
}

int itx_launch_finish(const itx_table *t, int mode, const uint64_t *p64, const uint32_t *p32, uint64_t *d_counts, uint32_t *d_cov,
                      uint32_t *d_cov_uniq, uint32_t *d_locus_out, hipStream_t st)
{
    const size_t n_counts = 2 * ((size_t)t->n_rep + t->n_fam + t->n_cla);
    ITX_HIP(hipMemsetAsync(d_counts, 0, (n_counts + 1) * sizeof(uint64_t), st));
    if (mode == ITX_MODE_STAT) {
        ITX_HIP(hipMemsetAsync(d_cov, 0, (t->cov_len + 1) * 4, st));
        ITX_HIP(hipMemsetAsync(d_cov_uniq, 0, (t->cov_len + 1) * 4, st));
        if (t->n_units) {
            hipLaunchKernelGGL(k_finish_unit, dim3(t->n_units), dim3(FB), 0, st, t->d_unit_slot, t->d_unit_ids, t->d_unit_covoff, p64,
                               p32, t->n_units, t->n_slots, t->n_rep, t->n_fam, t->n_cla, d_counts, d_cov, d_cov_uniq);
            ITX_HIP(hipGetLastError());
        }
    } else if (t->n_rows) {
        hipLaunchKernelGGL(k_permute_locus, dim3((t->n_rows + 255) / 256), dim3(256), 0, st, p32, t->dev.orig, t->n_rows, d_locus_out);
        ITX_HIP(hipGetLastError());
    }
    return ITX_OK;
}
