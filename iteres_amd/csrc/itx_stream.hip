// itx_stream.hip — the streaming kernel of the hot path: for every BAM record
//     derive (generic.c:748-905) -> classify (binRange.c:196-227 + generic.c:950-970) -> act,
// where "act" is, per template argument: nothing (classification only), global atomics into the A/B slot
// arrays (stat) or the per-locus counts (filter), or key emission for the partition path.
//
// Shape (gfx950): a workgroup owns one contiguous span of records; each of its 4 waves walks the span in
// tiles of 256 records, 4 CONSECUTIVE records per lane (16-byte loads of tid/pos/tmpend, 4-byte loads of
// mapq/flag5). For coordinate-sorted input a tile sits on one chromosome and a few kilobases, so per tile:
//   * the per-reference record (ItxTidRec) is a wave-uniform value cached across tiles,
//   * ONE coalesced load brings the slice of the binned index covering the tile into registers (lane j holds
//     bin j); the candidate window [lo_w, hi_w) of table rows comes out of it with two lane reads,
//   * ONE coalesced load stages the window's rows (32 B each, <= 128 rows) into the wave's LDS window,
//   * every record then finds its upper bound with two ds_bpermute reads of the index slice and a step or
//     two in LDS, and replays the reference's best-hit rule over LDS.
// Tiles that do not fit this picture (mixed chromosomes, > 64 bins or > 128 rows: unsorted or very sparse
// input) take the per-lane global-memory lookup — slower, same results.
// No MFMA: the path is integer compares and one f32 ratio; the roofline that bounds it is HBM.
#include "itx_device.h"

#define SB 256
#define RPL 4
#define WTILE (64 * RPL)

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

template <int WHAT>
__global__ __launch_bounds__(SB) void k_stream(ItxDevTable T, ItxRunParams P, ItxDevBatch B, size_t n, size_t span,
                                               int32_t *__restrict__ d_hit_row, uint64_t *__restrict__ u64,
                                               uint32_t *__restrict__ u32, ItxAccumLayout L, uint32_t *__restrict__ keys0,
                                               uint32_t *__restrict__ blk_cnt, ItxEmitPlan E)
{
    __shared__ uint4 s_win[SB / 64][2 * ITX_WIN];
    __shared__ uint32_t s_cnt[16];
    __shared__ uint32_t s_cursor;
    extern __shared__ uint32_t s_pc[];                         // EMIT: keys per partition of this workgroup's region
    if (threadIdx.x < 16) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_cursor = 0;
    if (WHAT == ITX_DO_EMIT)
        for (uint32_t k = threadIdx.x; k < E.n_part; k += SB) s_pc[k] = 0;
    __syncthreads();
    const uint32_t lane = lane_id();
    const uint32_t w = threadIdx.x >> 6;
    uint4 *win = s_win[w];
    const unsigned long long lt = (1ull << lane) - 1ull;
    const size_t begin = (size_t)blockIdx.x * span;
    size_t end = begin + span;
    if (end > n) end = n;
    uint32_t *out = keys0 ? keys0 + 2 * begin : nullptr;

    // wave-uniform cache of the current reference's ItxTidRec
    int32_t cur_tid = -0x7fffffff;
    uint4 cur0 = make_uint4(0xffffffffu, 0, 0, 0);
    uint32_t cur_bb = 0;
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0, c9 = 0, c10 = 0;

    for (size_t tb = begin + (size_t)w * WTILE; tb < end; tb += (size_t)(SB / 64) * WTILE) {
        const size_t r0 = tb + (size_t)lane * RPL;
        ItxRaw raw[RPL];
        bool ex[RPL];
        if (tb + WTILE <= end) {
            const int4 t4 = *reinterpret_cast<const int4 *>(B.tid + r0);
            const int4 p4 = *reinterpret_cast<const int4 *>(B.pos + r0);
            const int4 e4 = *reinterpret_cast<const int4 *>(B.tmpend + r0);
            const uint32_t mq = *reinterpret_cast<const uint32_t *>(B.mapq + r0);
            const uint32_t f4 = *reinterpret_cast<const uint32_t *>(B.flag5 + r0);
            raw[0] = {t4.x, p4.x, e4.x, mq & 0xffu, f4 & 0xffu};
            raw[1] = {t4.y, p4.y, e4.y, (mq >> 8) & 0xffu, (f4 >> 8) & 0xffu};
            raw[2] = {t4.z, p4.z, e4.z, (mq >> 16) & 0xffu, (f4 >> 16) & 0xffu};
            raw[3] = {t4.w, p4.w, e4.w, mq >> 24, f4 >> 24};
#pragma unroll
            for (int j = 0; j < RPL; j++) ex[j] = true;
        } else {
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                ex[j] = r0 + j < end;
                raw[j] = {0, 0, 0, 0, 0};
                if (ex[j]) raw[j] = {B.tid[r0 + j], B.pos[r0 + j], B.tmpend[r0 + j], B.mapq[r0 + j], B.flag5[r0 + j]};
            }
        }
        // ---- per-reference record: wave-uniform when every record of the tile shares one tid
        bool same = true;
#pragma unroll
        for (int j = 0; j < RPL; j++) same = same && (!ex[j] || raw[j].tid == cur_tid);
        if (__ballot(!same)) {
            const int32_t t0 = __shfl(raw[0].tid, 0, 64);          // record tb exists (tb < end)
            cur_tid = t0;
            if (t0 >= 0 && t0 < P.n_tid) {
                cur0 = *reinterpret_cast<const uint4 *>(&P.tidrec[t0]);
                cur_bb = P.tidrec[t0].bin_base;
            } else {
                cur0 = make_uint4(0xffffffffu, 0, 0, 0);
                cur_bb = 0;
            }
            same = true;
#pragma unroll
            for (int j = 0; j < RPL; j++) same = same && (!ex[j] || raw[j].tid == cur_tid);
        }
        const bool uniform = __ballot(!same) == 0ull;

        // ---- derive
        ItxDerived d[RPL];
        uint4 tr[RPL];
        uint32_t bb[RPL];
        int32_t qs[RPL], qe[RPL];
        bool q[RPL];
        bool anyq = false;
#pragma unroll
        for (int j = 0; j < RPL; j++) {
            tr[j] = cur0;
            bb[j] = cur_bb;
            if (!uniform) {
                const int32_t t = raw[j].tid;
                if (ex[j] && t >= 0 && t < P.n_tid) {
                    tr[j] = *reinterpret_cast<const uint4 *>(&P.tidrec[t]);
                    bb[j] = P.tidrec[t].bin_base;
                } else {
                    tr[j] = make_uint4(0xffffffffu, 0, 0, 0);
                }
            }
            d[j].cntbits = 0;
            d[j].ok = false;
            d[j].uniq = false;
            d[j].start = d[j].end = 0;
            if (ex[j]) d[j] = itx_derive(P, B, raw[j], tr[j], r0 + j);
            // binKeeperFind(bk, int start, int end) with its clipping (binRange.c:204-206)
            qs[j] = (int32_t)d[j].start;
            qe[j] = (int32_t)d[j].end;
            if (qs[j] < 0) qs[j] = 0;
            if (qe[j] > (int32_t)tr[j].y) qe[j] = (int32_t)tr[j].y;
            q[j] = d[j].ok && qs[j] < qe[j] && tr[j].z < tr[j].w;
            anyq = anyq || q[j];
        }

        // ---- classify
        int32_t hit[RPL];
        ItxIv rec[RPL];
#pragma unroll
        for (int j = 0; j < RPL; j++) hit[j] = -1;
        if (__ballot(anyq)) {
            bool fast = uniform;
            uint32_t lo_w = 0, wn = 0, bin_lo = 0;
            uint2 bs = make_uint2(0, 0);
            if (fast) {
                int32_t mn = 0x7fffffff, mx = 0;
#pragma unroll
                for (int j = 0; j < RPL; j++) {
                    if (q[j]) {
                        mn = qs[j] < mn ? qs[j] : mn;
                        mx = qe[j] > mx ? qe[j] : mx;
                    }
                }
                mn = wave_min_i32(mn);
                mx = wave_max_i32(mx);
                bin_lo = (uint32_t)mn >> T.shift;
                const uint32_t nb = ((uint32_t)mx >> T.shift) + 1 - bin_lo + 1;     // bins bin_lo .. bin(mx)+1
                fast = nb <= 64;
                if (fast) {
                    if (lane < nb) bs = T.bl[cur_bb + bin_lo + lane];
                    lo_w = (uint32_t)__shfl((int32_t)bs.y, 0, 64);
                    const uint32_t hi_w = (uint32_t)__shfl((int32_t)bs.x, (int)(nb - 1), 64);
                    wn = hi_w > lo_w ? hi_w - lo_w : 0u;
                    fast = wn <= ITX_WIN;
                }
            }
            if (fast) {
                if (wn) {
                    const uint4 *src = reinterpret_cast<const uint4 *>(T.iv + lo_w);
                    for (uint32_t k = lane; k < 2 * wn; k += 64) win[k] = src[k];       // coalesced, 16 B per lane
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    IvLds A{win, T.rank, lo_w};
#pragma unroll
                    for (int j = 0; j < RPL; j++) {
                        // upper bound: rows with s < qe end inside bin(qe); two lane reads of the index slice
                        const uint32_t b = q[j] ? ((uint32_t)qe[j] >> T.shift) - bin_lo : 0u;
                        uint32_t h0 = (uint32_t)__shfl((int32_t)bs.x, (int)b, 64);
                        uint32_t h1 = (uint32_t)__shfl((int32_t)bs.x, (int)b + 1, 64);
                        if (q[j]) {
                            h0 = h0 > lo_w ? h0 - lo_w : 0u;
                            h1 = h1 > lo_w ? h1 - lo_w : 0u;
                            uint32_t hi = h0;
                            if (h1 - h0 > 8) {
                                uint32_t a = h0, bnd = h1;
                                while (a < bnd) {
                                    const uint32_t m = (a + bnd) >> 1;
                                    if (A.s(m) < qe[j]) a = m + 1; else bnd = m;
                                }
                                hi = a;
                            } else {
                                while (hi < h1 && A.s(hi) < qe[j]) hi++;
                            }
                            const int32_t k = itx_pick(A, 0u, hi, qs[j], qe[j], d[j].start, d[j].end, P.min_cov);
                            if (k >= 0) {
                                hit[j] = (int32_t)lo_w + k;
                                const uint4 v0 = win[2 * k], v1 = win[2 * k + 1];
                                rec[j].s = (int32_t)v0.x; rec[j].e = (int32_t)v0.y; rec[j].pmax_e = (int32_t)v0.z; rec[j].cs = v0.w;
                                rec[j].jcap = v1.x; rec[j].covslot = v1.y; rec[j].zslot = v1.z; rec[j].unit = v1.w;
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();                                    // the window is rewritten next tile
                }
            } else {
#pragma unroll
                for (int j = 0; j < RPL; j++) {
                    if (q[j]) {
                        hit[j] = itx_classify_lane(T, tr[j].z, bb[j], qs[j], qe[j], d[j].start, d[j].end, P.min_cov);
                        if (hit[j] >= 0) rec[j] = T.iv[hit[j]];
                    }
                }
            }
        }

        // ---- cnt[] (generic.c:1048-1060): wave popcounts into scalar accumulators
#pragma unroll
        for (int j = 0; j < RPL; j++) {
            const uint32_t cb = d[j].cntbits;
            c0 += (uint32_t)__popcll(__ballot(cb & 1u));
            c1 += (uint32_t)__popcll(__ballot(cb & 2u));
            c2 += (uint32_t)__popcll(__ballot(cb & 4u));
            c3 += (uint32_t)__popcll(__ballot(cb & 8u));
            c4 += (uint32_t)__popcll(__ballot(cb & 16u));
            c5 += (uint32_t)__popcll(__ballot(cb & 32u));
            c6 += (uint32_t)__popcll(__ballot(cb & 64u));
            c7 += (uint32_t)__popcll(__ballot(cb & 128u));
            c9 += (uint32_t)__popcll(__ballot(hit[j] >= 0));                      // generic.c:1030-1032
            c10 += (uint32_t)__popcll(__ballot(hit[j] >= 0 && d[j].uniq));
        }

        // ---- chosen rows back to the caller (row ids as passed to itx_table_create)
        if (d_hit_row) {
            int32_t h[RPL];
#pragma unroll
            for (int j = 0; j < RPL; j++) h[j] = hit[j] >= 0 ? T.orig[hit[j]] : -1;
            if (tb + WTILE <= end) {
                *reinterpret_cast<int4 *>(d_hit_row + r0) = make_int4(h[0], h[1], h[2], h[3]);
            } else {
#pragma unroll
                for (int j = 0; j < RPL; j++)
                    if (ex[j]) d_hit_row[r0 + j] = h[j];
            }
        }

        // ---- act
        if (WHAT == ITX_DO_ATOMIC_STAT) {
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                if (hit[j] >= 0) {
                    uint32_t first;
                    const uint32_t nc = itx_cov_range(rec[j], d[j].start, d[j].end, &first);
                    if (nc) {
                        atomicAdd(&u32[L.a_all + first], 1u);
                        atomicAdd(&u32[L.b_all + first + nc], 1u);
                        if (d[j].uniq) {
                            atomicAdd(&u32[L.a_uniq + first], 1u);
                            atomicAdd(&u32[L.b_uniq + first + nc], 1u);
                        }
                    } else {
                        atomicAdd(&u32[L.a_all + rec[j].zslot], 1u);
                        if (d[j].uniq) atomicAdd(&u32[L.a_uniq + rec[j].zslot], 1u);
                    }
                }
            }
        } else if (WHAT == ITX_DO_ATOMIC_LOCUS) {
            // slCount(ss->sl) per locus (generic.c:662-666,1725): one atomic per run of equal rows in the wave
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                const int32_t h = hit[j];
                const int32_t hp = __shfl_up(h, 1, 64);
                const bool has = h >= 0;
                const bool st = has && (lane == 0 || hp != h);
                const unsigned long long m_st = __ballot(st), m_has = __ballot(has);
                if (st) {
                    // run = consecutive lanes with the same row: ends at the next run start or the next lane without a hit
                    const unsigned long long stop = (m_st | ~m_has) & ~((lt << 1) | 1ull);
                    const uint32_t e = stop ? (uint32_t)__ffsll((long long)stop) - 1u : 64u;
                    atomicAdd(&u32[L.locus + (uint32_t)h], e - lane);
                }
            }
        } else if (WHAT == ITX_DO_EMIT) {
            uint32_t kA[RPL], kB[RPL];
            bool hA[RPL], hB[RPL];
            uint32_t total = 0;
            unsigned long long mA[RPL], mB[RPL];
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                hA[j] = hit[j] >= 0;
                hB[j] = false;
                kA[j] = kB[j] = 0;
                if (hA[j]) {
                    uint32_t first;
                    const uint32_t nc = itx_cov_range(rec[j], d[j].start, d[j].end, &first);
                    const uint32_t u = d[j].uniq ? 1u : 0u;
                    if (nc) {
                        kA[j] = (first << 2) | u;
                        kB[j] = ((first + nc) << 2) | 2u | u;
                        hB[j] = true;
                    } else {
                        kA[j] = (rec[j].zslot << 2) | u;
                    }
                }
                mA[j] = __ballot(hA[j]);
                mB[j] = __ballot(hB[j]);
                total += (uint32_t)__popcll(mA[j]) + (uint32_t)__popcll(mB[j]);
            }
            uint32_t base = 0;
            if (lane == 0 && total) base = atomicAdd(&s_cursor, total);          // the workgroup's region cursor (LDS)
            base = (uint32_t)__shfl((int32_t)base, 0, 64);
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                if (hA[j]) out[base + (uint32_t)__popcll(mA[j] & lt)] = kA[j];
                base += (uint32_t)__popcll(mA[j]);
                if (hB[j]) out[base + (uint32_t)__popcll(mB[j] & lt)] = kB[j];
                base += (uint32_t)__popcll(mB[j]);
                // keys per partition: one LDS add per run of equal partitions (coordinate-sorted input: long runs)
                uint32_t len, leader;
                const uint32_t pA = kA[j] >> (2 + E.log_w), pB = kB[j] >> (2 + E.log_w);
                if (wave_run(pA, hA[j], lane, &len, &leader)) atomicAdd(&s_pc[pA], len);
                if (wave_run(pB, hB[j], lane, &len, &leader)) atomicAdd(&s_pc[pB], len);
            }
        }
    }
    if (WHAT != ITX_DO_CLASSIFY && lane == 0) {          // classify-only launches leave every accumulator alone
        if (c0) atomicAdd(&s_cnt[0], c0);
        if (c1) atomicAdd(&s_cnt[1], c1);
        if (c2) atomicAdd(&s_cnt[2], c2);
        if (c3) atomicAdd(&s_cnt[3], c3);
        if (c4) atomicAdd(&s_cnt[4], c4);
        if (c5) atomicAdd(&s_cnt[5], c5);
        if (c6) atomicAdd(&s_cnt[6], c6);
        if (c7) {
            atomicAdd(&s_cnt[7], c7);
            atomicAdd(&s_cnt[11], c7);                    // reads_nonredundant_unique == reads_mapped_unique without -R
        }
        if (c9) atomicAdd(&s_cnt[9], c9);
        if (c10) atomicAdd(&s_cnt[10], c10);
    }
    __syncthreads();
    if (WHAT != ITX_DO_CLASSIFY && threadIdx.x < 16 && s_cnt[threadIdx.x])
        atomicAdd((unsigned long long *)&u64[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    if (WHAT == ITX_DO_EMIT) {
        if (threadIdx.x == 0) blk_cnt[blockIdx.x] = s_cursor;
        // reserve this region's places: ONE add per touched partition, on one of 8 sub-cursors (workgroup id mod 8 —
        // workgroups are dealt round-robin over the 8 XCDs, so no address sees more than n_blocks/8 adds). The
        // offsets are kept as this region's row of the offset matrix for k_scatter.
        const uint32_t sub = blockIdx.x & (ITX_PART_SUB - 1);
        uint32_t *row = E.offm + (size_t)blockIdx.x * E.n_part;
        for (uint32_t k = threadIdx.x; k < E.n_part; k += SB) {
            const uint32_t c = s_pc[k];
            row[k] = c ? atomicAdd(&E.subcur[k * ITX_PART_SUB + sub], c) : 0u;
        }
    }
}

int itx_launch_stream(int what, const ItxDevTable &T, const ItxRunParams &P, const ItxDevBatch &B, size_t n, size_t span,
                      unsigned n_blocks, int32_t *d_hit_row, uint64_t *u64, uint32_t *u32, const ItxAccumLayout &L, uint32_t *keys0,
                      uint32_t *blk_cnt, const ItxEmitPlan &E, hipStream_t st)
{
    if (n == 0) return ITX_OK;
    const uintptr_t al = (uintptr_t)B.tid | (uintptr_t)B.pos | (uintptr_t)B.tmpend | (uintptr_t)d_hit_row;
    if ((al & 15u) || (((uintptr_t)B.mapq | (uintptr_t)B.flag5) & 3u)) {
        itx_set_error("record arrays must be 16-byte aligned (tid/pos/tmpend/hit_row) and 4-byte aligned (mapq/flag5)");
        return ITX_E_ARG;
    }
    if (span % ITX_STREAM_TILE) {
        itx_set_error("internal: span %zu is not a multiple of %u", span, ITX_STREAM_TILE);
        return ITX_E_ARG;
    }
    const dim3 g(n_blocks), b(SB);
    switch (what) {
    case ITX_DO_CLASSIFY:
        hipLaunchKernelGGL(k_stream<ITX_DO_CLASSIFY>, g, b, 0, st, T, P, B, n, span, d_hit_row, u64, u32, L, keys0, blk_cnt, E);
        break;
    case ITX_DO_ATOMIC_STAT:
        hipLaunchKernelGGL(k_stream<ITX_DO_ATOMIC_STAT>, g, b, 0, st, T, P, B, n, span, d_hit_row, u64, u32, L, keys0, blk_cnt, E);
        break;
    case ITX_DO_ATOMIC_LOCUS:
        hipLaunchKernelGGL(k_stream<ITX_DO_ATOMIC_LOCUS>, g, b, 0, st, T, P, B, n, span, d_hit_row, u64, u32, L, keys0, blk_cnt, E);
        break;
    case ITX_DO_EMIT:
        hipLaunchKernelGGL(k_stream<ITX_DO_EMIT>, g, b, (size_t)E.n_part * 4, st, T, P, B, n, span, d_hit_row, u64, u32, L, keys0,
                           blk_cnt, E);
        break;
    default:
        itx_set_error("internal: unknown stream action %d", what);
        return ITX_E_ARG;
    }
    ITX_HIP(hipGetLastError());
    return ITX_OK;
}
