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
//   * every record takes the top of its bin from the index slice (one ds_bpermute) and walks down the LDS
//     window while the prefix-max of the row ends still exceeds its start; the four records of a lane advance
//     in lockstep, one ds_read_b128 each per step.
// Tiles that do not fit this picture (mixed chromosomes, > 64 bins or > 128 rows: unsorted or very sparse
// input) take the per-lane global-memory lookup — slower, same results.
// The tile body is written predicated (selects, no early exits): the kernel is bound by instruction issue,
// and nested divergent branches cost more exec-mask bookkeeping than the arithmetic they skip.
// No MFMA: the path is integer compares and one f32 ratio; the roofline that bounds it is HBM.
#include "itx_device.h"

#define SB 256
#define RPL 4
#define WTILE (64 * RPL)

// generic.c:748-922 for one record, predicated. tr = first 16 bytes of the record's ItxTidRec.
struct Derived {
    uint32_t cntbits;      // bit k: cnt[k] += 1 for k in 0..7 (generic.c:1048-1055); cnt[11] == cnt[7] without -R
    uint32_t start, end;
    bool ok, uniq;
};
__device__ __forceinline__ Derived derive_pred(const ItxRunParams &P, const ItxRaw &r, const uint4 &tr, int32_t isz, int32_t mpos)
{
    Derived d;
    const uint32_t fl = r.fl;
    const bool paired = fl & F5_PAIRED, unmap = fl & F5_UNMAP, munmap = fl & F5_MUNMAP, rev = fl & F5_REVERSE, read1 = fl & F5_READ1;
    const bool treat = P.treat != 0;
    const bool end1 = !paired || read1 || treat;                                   // generic.c:748-759
    const bool mapped = !unmap;                                                    // generic.c:764
    const uint32_t cend = tr.y - 1u;                                               // generic.c:796
    const bool chrom_ok = mapped && (int32_t)tr.x >= 0 && cend != 1u;              // generic.c:781-801
    const bool se = treat || !paired || munmap;                                    // generic.c:815,836-837,885
    const uint32_t aisz = isz < 0 ? 0u - (uint32_t)isz : (uint32_t)isz;
    const bool pe_ok = read1 && aisz <= P.isize_max && isz != 0;                   // generic.c:838-840,858-860
    const bool se_ok = treat || !paired || P.discard == 0;                         // generic.c:862-863
    d.ok = chrom_ok && (se ? se_ok : pe_ok);
    d.uniq = r.mapq >= P.mapq_min;
    // generic.c:819-833
    uint32_t s_se = (uint32_t)r.pos;
    uint32_t e_se = umin32(cend, (uint32_t)r.tmpend);
    if (P.extension) {                                                             // wave-uniform
        const uint32_t e_plus = umin32(s_se + P.extension, cend);
        const uint32_t s_minus = e_se < P.extension ? 0u : e_se - P.extension;
        s_se = rev ? s_minus : s_se;
        e_se = rev ? e_se : e_plus;
    }
    // generic.c:845-855
    const bool fwd = isz > 0;
    const uint32_t s_pe = fwd ? (uint32_t)r.pos : (uint32_t)mpos;
    const uint32_t e_pe = umin32(cend, fwd ? s_pe + (uint32_t)isz : s_pe - (uint32_t)isz);
    d.start = se ? s_se : s_pe;
    d.end = se ? e_se : e_pe;
    d.cntbits = (end1 ? 1u : 2u) | (mapped ? (end1 ? 4u : 8u) : 0u) | (chrom_ok ? (end1 ? 16u : 32u) : 0u) | (d.ok ? 64u : 0u) |
                (d.ok && d.uniq ? 128u : 0u);
    return d;
}

// bit k of an 8-bit value -> 1 in nibble k
__device__ __forceinline__ uint32_t spread8(uint32_t x)
{
    uint32_t t = (x | (x << 12)) & 0x000f000fu;
    t = (t | (t << 3) | (t << 6) | (t << 9)) & 0x11111111u;
    return t;
}

__device__ __forceinline__ uint32_t uadd32(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    ITX_DPP_STEP(uadd32, v, 0, 0x111, 0xf);
    ITX_DPP_STEP(uadd32, v, 0, 0x112, 0xf);
    ITX_DPP_STEP(uadd32, v, 0, 0x114, 0xf);
    ITX_DPP_STEP(uadd32, v, 0, 0x118, 0xf);
    ITX_DPP_STEP(uadd32, v, 0, 0x142, 0xa);
    ITX_DPP_STEP(uadd32, v, 0, 0x143, 0xc);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

template <int WHAT>
__global__ __launch_bounds__(SB, 4) void k_stream(ItxDevTable T, ItxRunParams P, ItxDevBatch B, size_t n, size_t span,
                                               int32_t *__restrict__ d_hit_row, uint64_t *__restrict__ u64,
                                               uint32_t *__restrict__ u32, ItxAccumLayout L, uint32_t *__restrict__ keys0,
                                               uint32_t *__restrict__ blk_cnt, ItxEmitPlan E)
{
    __shared__ uint4 s_win[SB / 64][2 * ITX_WIN];
    __shared__ uint32_t s_lut[256];
    __shared__ uint32_t s_cnt[16];
    __shared__ uint32_t s_cursor;
    extern __shared__ uint32_t s_pc[];                         // EMIT: keys per partition of this workgroup's region
    s_lut[threadIdx.x] = spread8(threadIdx.x);
    if (threadIdx.x < 16) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_cursor = 0;
    if (WHAT == ITX_DO_EMIT)
        for (uint32_t k = threadIdx.x; k < E.n_part; k += SB) s_pc[k] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = threadIdx.x >> 6;
    uint4 *win = s_win[w];
    const unsigned long long lt = (1ull << lane) - 1ull;
    const size_t begin = (size_t)blockIdx.x * span;
    size_t end = begin + span;
    if (end > n) end = n;
    uint32_t *out = keys0 ? keys0 + 2 * begin : nullptr;
    const bool have_pe = B.isize != nullptr;

    // wave-uniform cache of the current reference's ItxTidRec
    int32_t cur_tid = -0x7fffffff;
    uint4 cur0 = make_uint4(0xffffffffu, 0, 0, 0);
    uint32_t cur_bb = 0;
    // per-lane counters (generic.c:1048-1060), reduced over the wave once at the end
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, acc_hit = 0, acc_hitu = 0;
#ifdef ITX_ABLATE
    uint32_t sink = 0;      // timing-only builds: stop the tile early, keep what was computed alive
#define ITX_ABLATE_AT(k, expr) if (ITX_ABLATE == (k)) { sink += (expr); continue; }
#else
#define ITX_ABLATE_AT(k, expr)
#endif

    for (size_t tb = begin + (size_t)w * WTILE; tb < end; tb += (size_t)(SB / 64) * WTILE) {
        const size_t r0 = tb + (size_t)lane * RPL;
        const bool full = tb + WTILE <= end;                                    // wave-uniform
        ItxRaw raw[RPL];
        bool ex[RPL];
        int32_t isz[RPL] = {0, 0, 0, 0}, mps[RPL] = {0, 0, 0, 0};
        if (full) {
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
            if (have_pe && __ballot(f4 & 0x01010101u)) {                        // some record of the tile is paired
                const int4 i4 = *reinterpret_cast<const int4 *>(B.isize + r0);
                const int4 m4 = *reinterpret_cast<const int4 *>(B.mpos + r0);
                isz[0] = i4.x; isz[1] = i4.y; isz[2] = i4.z; isz[3] = i4.w;
                mps[0] = m4.x; mps[1] = m4.y; mps[2] = m4.z; mps[3] = m4.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                ex[j] = r0 + j < end;
                raw[j] = {0, 0, 0, 0, 0};
                if (ex[j]) {
                    raw[j] = {B.tid[r0 + j], B.pos[r0 + j], B.tmpend[r0 + j], B.mapq[r0 + j], B.flag5[r0 + j]};
                    if (have_pe) {
                        isz[j] = B.isize[r0 + j];
                        mps[j] = B.mpos[r0 + j];
                    }
                }
            }
        }
        ITX_ABLATE_AT(1, raw[0].tid + raw[1].pos + raw[2].tmpend + raw[3].mapq + raw[3].fl)
        // ---- per-reference record: wave-uniform when every record of the tile shares one tid
        bool same = true;
#pragma unroll
        for (int j = 0; j < RPL; j++) same = same && (!ex[j] || raw[j].tid == cur_tid);
        if (__ballot(!same)) {
            const int32_t t0 = __builtin_amdgcn_readfirstlane(raw[0].tid);     // record tb exists (tb < end)
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
        Derived d[RPL];
        uint4 tr[RPL];
        uint32_t bb[RPL];
        int32_t qs[RPL], qe[RPL];
        bool q[RPL];
        bool anyq = false;
#pragma unroll
        for (int j = 0; j < RPL; j++) {
            tr[j] = cur0;
            bb[j] = cur_bb;
        }
        if (!uniform) {                                                         // mixed references in one tile: rare
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                const int32_t t = raw[j].tid;
                if (ex[j] && t >= 0 && t < P.n_tid) {
                    tr[j] = *reinterpret_cast<const uint4 *>(&P.tidrec[t]);
                    bb[j] = P.tidrec[t].bin_base;
                } else {
                    tr[j] = make_uint4(0xffffffffu, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < RPL; j++) {
            d[j] = derive_pred(P, raw[j], tr[j], isz[j], mps[j]);
            if (!ex[j]) {
                d[j].cntbits = 0;
                d[j].ok = false;
            }
            // binKeeperFind(bk, int start, int end) with its clipping (binRange.c:204-206)
            qs[j] = imax32((int32_t)d[j].start, 0);
            qe[j] = imin32((int32_t)d[j].end, (int32_t)tr[j].y);
            q[j] = d[j].ok && qs[j] < qe[j] && tr[j].z < tr[j].w;
            anyq = anyq || q[j];
        }
        ITX_ABLATE_AT(2, (uint32_t)qs[0] + (uint32_t)qe[1] + (uint32_t)qs[2] + (uint32_t)qe[3] + d[0].cntbits + d[1].cntbits + d[2].cntbits + d[3].cntbits)

        // ---- classify
        int32_t hit[RPL] = {-1, -1, -1, -1};
        ItxIv rec[RPL];
#pragma unroll
        for (int j = 0; j < RPL; j++) rec[j] = ItxIv{0, 0, 0, 0, 0, 0, 0, 0};
        if (__ballot(anyq)) {
            bool fast = uniform;
            uint32_t lo_w = 0, wn = 0, bin_lo = 0;
            uint2 bs = make_uint2(0, 0);
            if (fast) {
                int32_t mn = 0x7fffffff, mx = 0;
#pragma unroll
                for (int j = 0; j < RPL; j++) {
                    mn = q[j] ? imin32(qs[j], mn) : mn;
                    mx = q[j] ? imax32(qe[j], mx) : mx;
                }
                mn = wave_min_i32(mn);
                mx = wave_max_i32(mx);
                bin_lo = (uint32_t)mn >> T.shift;
                const uint32_t nb = ((uint32_t)mx >> T.shift) + 1 - bin_lo + 1;     // bins bin_lo .. bin(mx)+1
                fast = nb <= 64;
                if (fast) {
                    if (lane < nb) bs = T.bl[cur_bb + bin_lo + lane];
                    lo_w = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)bs.y);
                    const uint32_t hi_w = (uint32_t)__builtin_amdgcn_readlane((int32_t)bs.x, (int)(nb - 1));
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
                    ITX_ABLATE_AT(3, win[2 * (lane % wn)].x + wn)
                    // Candidates of a record: window rows below the top of qe's bin (one lane read of the index
                    // slice); rows starting at or after qe fail the overlap test by themselves. The four records of a
                    // lane walk down in lockstep while the prefix-max of the ends still exceeds qs
                    // (binRange.c:209-225 without the bin lists).
                    uint32_t kk[RPL], top[RPL], low[RPL];
                    uint32_t h_k[RPL], h_ov[RPL], h_rk[RPL];      // the hit found last (lowest row so far)
                    uint32_t g_k[RPL], g_ov[RPL], g_rk[RPL];      // the hit found before it
                    int32_t nh[RPL];
                    bool act[RPL];
                    bool any = false;
#pragma unroll
                    for (int j = 0; j < RPL; j++) {
                        const uint32_t b = q[j] ? ((uint32_t)qe[j] >> T.shift) - bin_lo + 1u : 0u;
                        uint32_t h1 = (uint32_t)__shfl((int32_t)bs.x, (int)b, 64);
                        h1 = h1 > lo_w ? h1 - lo_w : 0u;
                        top[j] = kk[j] = low[j] = h1;
                        act[j] = q[j] && h1 > 0;
                        nh[j] = 0;
                        h_k[j] = h_ov[j] = h_rk[j] = g_k[j] = g_ov[j] = g_rk[j] = 0;
                        any = any || act[j];
                    }
                    while (any) {
                        uint4 v[RPL];
#pragma unroll
                        for (int j = 0; j < RPL; j++) {
                            kk[j] = act[j] ? kk[j] - 1 : 0u;
                            v[j] = win[2 * kk[j]];                                      // s, e, pmax_e, rank
                        }
                        any = false;
#pragma unroll
                        for (int j = 0; j < RPL; j++) {
                            const bool alive = act[j] && (int32_t)v[j].z > qs[j];
                            const int32_t ov = clip_ov((int32_t)v[j].x, (int32_t)v[j].y, qs[j], qe[j]);
                            const bool ovl = alive && ov > 0;
                            nh[j] += ovl ? 1 : 0;
                            g_k[j] = ovl ? h_k[j] : g_k[j];
                            g_ov[j] = ovl ? h_ov[j] : g_ov[j];
                            g_rk[j] = ovl ? h_rk[j] : g_rk[j];
                            h_k[j] = ovl ? kk[j] : h_k[j];
                            h_ov[j] = ovl ? (uint32_t)ov : h_ov[j];
                            h_rk[j] = ovl ? v[j].w : h_rk[j];
                            low[j] = alive ? kk[j] : low[j];
                            act[j] = alive && kk[j] > 0;
                            any = any || act[j];
                        }
                    }
                    // Best hit (generic.c:950-970): in binKeeperFind's list order, the LAST hit whose coverage exceeds the
                    // previous hit's. All hits of a record share the denominator (end - start) and, for overlaps below
                    // 2^23, distinct integer overlaps give distinct f32 quotients — so with two hits the pick is an integer
                    // comparison on (rank, overlap); three or more hits (or giant fragments) replay the rule in full.
                    bool multi = false;
#pragma unroll
                    for (int j = 0; j < RPL; j++) {
                        const bool h_first = h_rk[j] < g_rk[j];                        // which of the two comes first in list order
                        const uint32_t ov1 = h_first ? h_ov[j] : g_ov[j], ov2 = h_first ? g_ov[j] : h_ov[j];
                        const uint32_t k1 = h_first ? h_k[j] : g_k[j], k2 = h_first ? g_k[j] : h_k[j];
                        const bool two = nh[j] == 2;
                        const bool second = two && ov2 > ov1;
                        const uint32_t ck = two ? (second ? k2 : k1) : h_k[j];
                        const uint32_t cov_ov = two ? (second ? ov2 : ov1) : h_ov[j];
                        const uint32_t qlen = d[j].end - d[j].start;
                        const float c = __fdiv_rn((float)cov_ov, (float)qlen);          // generic.c:296-301 (qlen > 0 for any hit)
                        const bool big = qlen >= (1u << 23);
                        const bool simple = (nh[j] == 1 || two) && !big;
                        hit[j] = (simple && !(c < P.min_cov)) ? (int32_t)ck : -1;       // generic.c:961-962
                        multi = multi || (nh[j] > 0 && !simple);
                    }
                    if (__ballot(multi)) {                                              // three or more hits / giant fragments: rare
                        IvLds A{win};
#pragma unroll
                        for (int j = 0; j < RPL; j++) {
                            const bool simple = (nh[j] == 1 || nh[j] == 2) && (d[j].end - d[j].start) < (1u << 23);
                            if (nh[j] > 0 && !simple)
                                hit[j] = itx_pick_multi(A, low[j], top[j], qs[j], qe[j], d[j].start, d[j].end, P.min_cov);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < RPL; j++) {
                        const uint32_t k = hit[j] >= 0 ? (uint32_t)hit[j] : 0u;
                        const uint4 v0 = win[2 * k], v1 = win[2 * k + 1];
                        rec[j].s = (int32_t)v0.x; rec[j].e = (int32_t)v0.y; rec[j].pmax_e = (int32_t)v0.z; rec[j].rank = v0.w;
                        rec[j].cs = v1.x; rec[j].jcap = v1.y; rec[j].covslot = v1.z; rec[j].zslot = v1.w;
                        hit[j] = hit[j] >= 0 ? hit[j] + (int32_t)lo_w : -1;
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
        ITX_ABLATE_AT(4, (uint32_t)(hit[0] + hit[1] + hit[2] + hit[3]))

        // ---- cnt[] (generic.c:1048-1060): nibble-wise per-lane sums of the four records' bits
        {
            const uint32_t n4 = s_lut[d[0].cntbits] + s_lut[d[1].cntbits] + s_lut[d[2].cntbits] + s_lut[d[3].cntbits];
#pragma unroll
            for (int k = 0; k < 8; k++) acc[k] += (n4 >> (4 * k)) & 0xfu;
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                acc_hit += hit[j] >= 0 ? 1u : 0u;                                       // generic.c:1030-1032
                acc_hitu += (hit[j] >= 0 && d[j].uniq) ? 1u : 0u;
            }
        }
        ITX_ABLATE_AT(5, (uint32_t)(hit[0] + hit[1] + hit[2] + hit[3]) + acc[0] + acc_hit)

        // ---- chosen rows back to the caller (row ids as passed to itx_table_create)
        if (d_hit_row) {
            int32_t h[RPL];
#pragma unroll
            for (int j = 0; j < RPL; j++) h[j] = hit[j] >= 0 ? T.orig[hit[j]] : -1;
            if (full) {
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
                uint32_t len, leader;
                if (wave_run((uint32_t)hit[j], hit[j] >= 0, lane, &len, &leader)) atomicAdd(&u32[L.locus + (uint32_t)hit[j]], len);
            }
        } else if (WHAT == ITX_DO_EMIT) {
            uint32_t kA[RPL], kB[RPL];
            bool hA[RPL], hB[RPL];
            uint32_t total = 0;
            unsigned long long mA[RPL], mB[RPL];
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                uint32_t first;
                const uint32_t nc = itx_cov_range(rec[j], d[j].start, d[j].end, &first);
                const uint32_t u = d[j].uniq ? 1u : 0u;
                hA[j] = hit[j] >= 0;
                hB[j] = hA[j] && nc != 0;
                kA[j] = ((hB[j] ? first : rec[j].zslot) << 2) | u;                     // no coverage: one start in the unit's extra slot
                kB[j] = ((first + nc) << 2) | 2u | u;
                mA[j] = __ballot(hA[j]);
                mB[j] = __ballot(hB[j]);
                total += (uint32_t)__popcll(mA[j]) + (uint32_t)__popcll(mB[j]);
            }
            uint32_t base = 0;
            if (lane == 0 && total) base = atomicAdd(&s_cursor, total);          // the workgroup's region cursor (LDS)
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)base);
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                if (hA[j]) out[base + (uint32_t)__popcll(mA[j] & lt)] = kA[j];
                base += (uint32_t)__popcll(mA[j]);
                if (hB[j]) out[base + (uint32_t)__popcll(mB[j] & lt)] = kB[j];
                base += (uint32_t)__popcll(mB[j]);
                // keys per partition of this workgroup's region (LDS adds to equal addresses serialise in the LDS
                // pipeline, beside the VALU work, which is what bounds this kernel)
                if (hA[j]) atomicAdd(&s_pc[kA[j] >> (2 + E.log_w)], 1u);
                if (hB[j]) atomicAdd(&s_pc[kB[j] >> (2 + E.log_w)], 1u);
            }
        }
    }
#ifdef ITX_ABLATE
    if (sink == 0x7fffff01u) s_cnt[12] = sink;
#endif
    if (WHAT != ITX_DO_CLASSIFY) {                        // classify-only launches leave every accumulator alone
        uint32_t tot[10];
#pragma unroll
        for (int k = 0; k < 8; k++) tot[k] = wave_sum_u32(acc[k]);
        tot[8] = wave_sum_u32(acc_hit);
        tot[9] = wave_sum_u32(acc_hitu);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (tot[k]) atomicAdd(&s_cnt[k], tot[k]);
            if (tot[7]) atomicAdd(&s_cnt[11], tot[7]);    // reads_nonredundant_unique == reads_mapped_unique without -R
            if (tot[8]) atomicAdd(&s_cnt[9], tot[8]);
            if (tot[9]) atomicAdd(&s_cnt[10], tot[9]);
        }
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
    const uintptr_t al = (uintptr_t)B.tid | (uintptr_t)B.pos | (uintptr_t)B.tmpend | (uintptr_t)d_hit_row | (uintptr_t)B.mpos | (uintptr_t)B.isize;
    if ((al & 15u) || (((uintptr_t)B.mapq | (uintptr_t)B.flag5) & 3u)) {
        itx_set_error("record arrays must be 16-byte aligned (tid/pos/tmpend/mpos/isize/hit_row) and 4-byte aligned (mapq/flag5)");
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
