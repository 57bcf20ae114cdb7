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
//     window while the rows below still end past its start (ItxIv.pbelow); the four records of a lane advance
//     in lockstep, one ds_read_b128 each per step.
// Tiles that do not fit this picture (mixed chromosomes, > 128 bins or > 128 rows: unsorted or very sparse
// input) take the per-lane global-memory lookup — slower, same results.
// The tile body is written predicated (selects, no early exits): the kernel is bound by instruction issue,
// and nested divergent branches cost more exec-mask bookkeeping than the arithmetic they skip.
// No MFMA: the path is integer compares and one f32 ratio; the roofline that bounds it is HBM.
#include "itx_device.h"
#include <stdlib.h>

#define SB 256
#ifndef ITX_LB
#define ITX_LB ITX_STREAM_LB
#endif
#ifndef ITX_RPL
#define ITX_RPL 4          // records per lane and tile: 4 (16-byte loads) or 2 (8-byte loads, half the per-wave state)
#endif
#define RPL ITX_RPL
#define WTILE (64 * RPL)

// Everything generic.c:748-922 decides from a record's flag bits alone, tabulated once per workgroup.
// Index: flag5 (6 bits) | 64 the reference is known and usable (generic.c:781-801) | 128 a proper-pair insert size
// (generic.c:838-840) | 256 MAPQ >= -Q. Entry: bit 3k set => cnt[k] += 1 for k in 0..7 (generic.c:1048-1055;
// cnt[11] == cnt[7] without -R), LUT_OK the record goes on to the lookup, LUT_SE it is measured as a single end.
#define LUT_OK (1u << 24)
#define LUT_SE (1u << 25)
__device__ __forceinline__ uint32_t lut_entry(const ItxRunParams &P, uint32_t idx)
{
    const bool paired = idx & F5_PAIRED, unmap = idx & F5_UNMAP, munmap = idx & F5_MUNMAP, read1 = idx & F5_READ1;
    const bool ref_ok = idx & 64u, isz_ok = idx & 128u, uniq = idx & 256u;
    const bool treat = P.treat != 0;
    const bool end1 = !paired || read1 || treat;                                   // generic.c:748-759
    const bool mapped = !unmap;                                                    // generic.c:764
    const bool chrom_ok = mapped && ref_ok;                                        // generic.c:781-801
    const bool se = treat || !paired || munmap;                                    // generic.c:815,836-837,885
    const bool pe_ok = read1 && isz_ok;                                            // generic.c:838-840,858-860
    const bool se_ok = treat || !paired || P.discard == 0;                         // generic.c:862-863
    const bool ok = chrom_ok && (se ? se_ok : pe_ok);
    uint32_t e = end1 ? 1u : 1u << 3;
    e |= mapped ? (end1 ? 1u << 6 : 1u << 9) : 0u;
    e |= chrom_ok ? (end1 ? 1u << 12 : 1u << 15) : 0u;
    e |= ok ? 1u << 18 : 0u;
    e |= (ok && uniq) ? 1u << 21 : 0u;
    return e | ((ok && !(idx & F5_NOLOOKUP)) ? LUT_OK : 0u) | (se ? LUT_SE : 0u);      // the caller's -R / XA `continue`
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

// Top of a record's candidates as a window entry: the first row starting at or after the end of qe's bin, from
// the wave's slice of the binned index (lane i holds bin bin_lo + i in bsx and, for slices of more than 64 bins —
// sparser read sets — bin bin_lo + 64 + i in bsx_hi), relative to the window start.
__device__ __forceinline__ uint32_t top_entry(uint32_t bsx, uint32_t bsx_hi, bool wide, int32_t qe, int32_t shift, uint32_t bin_lo, uint32_t lo_w)
{
    const uint32_t b = (((uint32_t)qe >> shift) - bin_lo + 1u) & 127u;

    uint32_t h1 = (uint32_t)__shfl((int32_t)bsx, (int)(b & 63u), 64);
    if (wide) {                                                    // wave-uniform
        const uint32_t h2 = (uint32_t)__shfl((int32_t)bsx_hi, (int)(b & 63u), 64);
        h1 = b >= 64u ? h2 : h1;
    }
    return __builtin_elementwise_sub_sat(h1, lo_w);
}

// generic.c:748-922 for one record. (tx, ty) = chrom and size of the record's ItxTidRec, has_rows = its reference
// has table rows. Out: the flag table's entry, the reference's unsigned start/end, binKeeperFind's clipped query
// (binRange.c:204-206), whether the record goes on to the lookup, and MAPQ >= -Q.
__device__ __forceinline__ void derive_one(const ItxRunParams &P, const uint32_t *s_lut, const ItxRaw &r, int32_t iz, int32_t mpos, bool tile_pe,
                                           uint32_t tx, uint32_t ty, bool has_rows, uint32_t &lut, uint32_t &st, uint32_t &en, int32_t &qs,
                                           int32_t &qe, bool &q, bool &uq)
{
    const uint32_t cend = ty - 1u;                                                 // generic.c:796
    uq = r.mapq >= P.mapq_min;
    uint32_t idx = r.fl | (((int32_t)tx >= 0 && cend != 1u) ? 64u : 0u) | (uq ? 256u : 0u);
    // generic.c:819-833
    uint32_t s_se = (uint32_t)r.pos;
    uint32_t e_se = umin32(cend, (uint32_t)r.tmpend);
    if (P.extension) {                                                             // wave-uniform
        const bool rev = r.fl & F5_REVERSE;
        const uint32_t e_plus = umin32(s_se + P.extension, cend);
        const uint32_t s_minus = __builtin_elementwise_sub_sat(e_se, P.extension);
        s_se = rev ? s_minus : s_se;
        e_se = rev ? e_se : e_plus;
    }
    st = s_se;
    en = e_se;
    if (tile_pe) {                                                                 // wave-uniform; generic.c:838-855
        const uint32_t aisz = iz < 0 ? 0u - (uint32_t)iz : (uint32_t)iz;
        idx |= (aisz <= P.isize_max && iz != 0) ? 128u : 0u;
        lut = s_lut[idx];
        const bool se = lut & LUT_SE;
        const bool fwd = iz > 0;
        const uint32_t s_pe = fwd ? (uint32_t)r.pos : (uint32_t)mpos;
        const uint32_t e_pe = umin32(cend, fwd ? s_pe + (uint32_t)iz : s_pe - (uint32_t)iz);
        st = se ? s_se : s_pe;
        en = se ? e_se : e_pe;
    } else {
        lut = s_lut[idx];
    }
    qs = imax32((int32_t)st, 0);
    qe = imin32((int32_t)en, (int32_t)ty);
    q = (lut & LUT_OK) && qs < qe && has_rows;
}

// One record classified straight from global memory (any record order), and the slots its chosen row marks.
__device__ __forceinline__ void classify_global(const ItxDevTable &T, const ItxRunParams &P, uint32_t iv_lo, uint32_t bin_base, int32_t qs, int32_t qe,
                                                uint32_t st, uint32_t en, int32_t &hit, uint32_t &sA, uint32_t &sB, bool &hB)
{
    hit = itx_classify_lane(T, iv_lo, bin_base, qs, qe, st, en, P.min_cov);
    if (hit >= 0) {
        const ItxIv r = T.iv[hit];
        uint32_t first;
        const uint32_t nc = itx_cov_range(r, st, en, &first);
        hB = nc != 0;
        sA = hB ? first : r.zslot;
        sB = first + nc;
    }
}

template <int WHAT>
__global__ __launch_bounds__(SB, ITX_LB) void k_stream(ItxDevTable T, ItxRunParams P, ItxDevBatch B, size_t n, size_t span,
                                               int32_t *__restrict__ d_hit_row, uint64_t *__restrict__ u64,
                                               uint32_t *__restrict__ u32, ItxAccumLayout L, uint2 *__restrict__ keys0,
                                               uint32_t *__restrict__ blk_cnt, ItxEmitPlan E)
{
    // a wave's window: entry 0 is a sentinel no query overlaps and no scan walks past, table row lo_w + i sits at entry i + 1
    __shared__ uint4 s_win[SB / 64][2 * (ITX_WIN + 1)];
    __shared__ uint32_t s_lut[1024];                           // entries 512.. are zero: records past the end carry fl = 512
    __shared__ uint32_t s_cnt[16];
    extern __shared__ uint32_t s_pc[];                         // EMIT: keys per partition of this workgroup's region
    constexpr bool FIRST = WHAT == ITX_DO_FIND_FIRST;         // cpg lookups: first hit in list order, plain intervals
    constexpr bool EMIT = WHAT == ITX_DO_EMIT || WHAT == ITX_DO_EMIT_WIDE;
    constexpr bool WIDE = WHAT == ITX_DO_EMIT_WIDE;           // partitions wider than k_hist's LDS window (slot spaces beyond 33.5 M)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = threadIdx.x >> 6;
    uint4 *win = s_win[w];
    s_lut[threadIdx.x] = lut_entry(P, threadIdx.x);
    s_lut[256 + threadIdx.x] = lut_entry(P, 256 + threadIdx.x);
    s_lut[512 + threadIdx.x] = 0;
    s_lut[768 + threadIdx.x] = 0;
    if (threadIdx.x < 16) s_cnt[threadIdx.x] = 0;
    if (lane == 0) {
        win[0] = make_uint4(0x3fffffffu, 0xc0000000u, 0x80000000u, 0xffffffffu);      // s, e, pbelow, rank
        win[1] = make_uint4(0, 0, 0, 0);
    }
    if (EMIT)
        for (uint32_t k = threadIdx.x; k < E.n_part; k += SB) s_pc[k] = 0;
    __syncthreads();
    const size_t begin = (size_t)blockIdx.x * span;
    size_t end = begin + span;
    if (end > n) end = n;
    uint2 *w_out = keys0 ? keys0 + 2 * begin + (size_t)w * (span / 2) : nullptr;    // the wave's quarter: at most two keys per record
    uint32_t w_keys = 0;
    const bool have_pe = B.isize != nullptr;

    // wave-uniform cache of the current reference's ItxTidRec
    int32_t cur_tid = -0x7fffffff;
    uint4 cur0 = make_uint4(0xffffffffu, 0, 0, 0);
    uint32_t cur_bb = 0;
    // cnt[0..7] (generic.c:1048-1055): per-lane sums in 6-bit fields (even counters in accA, odd ones in accB), spilled
    // into 16-bit halves before a field can overflow (itx_launch_stream bounds the tiles per wave); hits (generic.c:1030-1032) are counted per wave from ballots
    uint32_t accA = 0, accB = 0, acc[4] = {0, 0, 0, 0};        // acc[k]: cnt[2k] | cnt[2k+1] << 16
    uint32_t n_hit = 0, n_hitu = 0, tiles = 0;
#ifdef ITX_ABLATE
    uint32_t sink = 0;      // timing-only builds: stop the tile early, keep what was computed alive
#define ITX_ABLATE_AT(k, expr) if (ITX_ABLATE == (k)) { sink += (expr); continue; }
#define ITX_ABLATE_NOT(k) (ITX_ABLATE != (k))     // timing-only builds that leave ONE piece out (6: partition counts, 7: key stores)
#else
#define ITX_ABLATE_AT(k, expr)
#define ITX_ABLATE_NOT(k) true
#endif

    for (size_t tb = begin + (size_t)w * WTILE; tb < end; tb += (size_t)(SB / 64) * WTILE) {
        const size_t r0 = tb + (size_t)lane * RPL;
        const bool full = tb + WTILE <= end;                                    // wave-uniform
        ItxRaw raw[RPL];
        bool ex[RPL];
        int32_t isz[RPL], mps[RPL];
#pragma unroll
        for (int j = 0; j < RPL; j++) isz[j] = mps[j] = 0;
        bool tile_pe = have_pe;                                                 // wave-uniform: mate fields were read
        if (full) {
#if RPL == 4
#ifndef ITX_NO_NT
            // the records go by once: streaming loads leave the caches to the table (measured: 1 % at 500 M records)
            typedef int v4i __attribute__((ext_vector_type(4)));
            const v4i t4v = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(B.tid + r0));
            const v4i p4v = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(B.pos + r0));
            const v4i e4v = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(B.tmpend + r0));
            const int4 t4 = make_int4(t4v.x, t4v.y, t4v.z, t4v.w), p4 = make_int4(p4v.x, p4v.y, p4v.z, p4v.w), e4 = make_int4(e4v.x, e4v.y, e4v.z, e4v.w);
            const uint32_t mq = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(B.mapq + r0));
            const uint32_t f4 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(B.flag5 + r0));
#else
            const int4 t4 = *reinterpret_cast<const int4 *>(B.tid + r0);
            const int4 p4 = *reinterpret_cast<const int4 *>(B.pos + r0);
            const int4 e4 = *reinterpret_cast<const int4 *>(B.tmpend + r0);
            const uint32_t mq = *reinterpret_cast<const uint32_t *>(B.mapq + r0);
            const uint32_t f4 = *reinterpret_cast<const uint32_t *>(B.flag5 + r0);
#endif
            raw[0] = {t4.x, p4.x, e4.x, mq & 0xffu, f4 & 0x3fu};
            raw[1] = {t4.y, p4.y, e4.y, (mq >> 8) & 0xffu, (f4 >> 8) & 0x3fu};
            raw[2] = {t4.z, p4.z, e4.z, (mq >> 16) & 0xffu, (f4 >> 16) & 0x3fu};
            raw[3] = {t4.w, p4.w, e4.w, mq >> 24, (f4 >> 24) & 0x3fu};
            const uint32_t paired_mask = 0x01010101u;
#else
            const int2 t4 = *reinterpret_cast<const int2 *>(B.tid + r0);
            const int2 p4 = *reinterpret_cast<const int2 *>(B.pos + r0);
            const int2 e4 = *reinterpret_cast<const int2 *>(B.tmpend + r0);
            const uint32_t mq = *reinterpret_cast<const uint16_t *>(B.mapq + r0);
            const uint32_t f4 = *reinterpret_cast<const uint16_t *>(B.flag5 + r0);
            raw[0] = {t4.x, p4.x, e4.x, mq & 0xffu, f4 & 0x3fu};
            raw[1] = {t4.y, p4.y, e4.y, (mq >> 8) & 0xffu, (f4 >> 8) & 0x3fu};
            const uint32_t paired_mask = 0x0101u;
#endif
#pragma unroll
            for (int j = 0; j < RPL; j++) ex[j] = true;
            tile_pe = have_pe && __ballot(f4 & paired_mask) != 0ull;            // some record of the tile is paired
            if (tile_pe) {
#if RPL == 4
                const int4 i4 = *reinterpret_cast<const int4 *>(B.isize + r0);
                const int4 m4 = *reinterpret_cast<const int4 *>(B.mpos + r0);
                isz[0] = i4.x; isz[1] = i4.y; isz[2] = i4.z; isz[3] = i4.w;
                mps[0] = m4.x; mps[1] = m4.y; mps[2] = m4.z; mps[3] = m4.w;
#else
                const int2 i4 = *reinterpret_cast<const int2 *>(B.isize + r0);
                const int2 m4 = *reinterpret_cast<const int2 *>(B.mpos + r0);
                isz[0] = i4.x; isz[1] = i4.y;
                mps[0] = m4.x; mps[1] = m4.y;
#endif
            }
        } else {
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                ex[j] = r0 + j < end;
                raw[j] = {0, 0, 0, 0, 512u};
                if (ex[j]) {
                    raw[j] = {B.tid[r0 + j], B.pos[r0 + j], B.tmpend[r0 + j], B.mapq[r0 + j], B.flag5[r0 + j] & 0x3fu};
                    if (have_pe) {
                        isz[j] = B.isize[r0 + j];
                        mps[j] = B.mpos[r0 + j];
                    }
                }
            }
        }
        ITX_ABLATE_AT(1, raw[0].tid + raw[1].pos + raw[0].tmpend + raw[1].mapq + raw[1].fl)
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

        // ---- derive (generic.c:748-922): flag logic from the table, coordinates predicated, the reference's record in
        // scalar registers. A tile that straddles references (a handful per sorted file), or holds a record whose
        // unsigned start/end differ from the clipped query binKeeperFind sees (negative positions), is taken record by
        // record straight away — own reference record, lookup in global memory — while the raw fields are at hand.
        uint32_t lut[RPL];
        int32_t qs[RPL], qe[RPL];                  // for every record that goes on: also the reference's start / end
        bool q[RPL], uq[RPL];
        bool anyq = false, odd = !uniform;
        int32_t hit[RPL];
        uint32_t sA[RPL], sB[RPL];                 // slots of the start / end marks of the chosen row's consensus range
        bool hB[RPL];                              // the read adds coverage (else: one start in the unit's extra slot)
#pragma unroll
        for (int j = 0; j < RPL; j++) {
            hit[j] = -1;
            sA[j] = sB[j] = 0;
            hB[j] = false;
        }
#pragma unroll
        for (int j = 0; j < RPL; j++) {
            if (FIRST) {
                // a plain interval [pos, tmpend) on a chromosome with rows, clipped like binKeeperFind (binRange.c:204-206)
                lut[j] = 0;
                uq[j] = false;
                qs[j] = imax32(raw[j].pos, 0);
                qe[j] = imin32(raw[j].tmpend, (int32_t)cur0.y);
                q[j] = raw[j].fl != 512u && (int32_t)cur0.x >= 0 && cur0.z < cur0.w && qs[j] < qe[j];
            } else {
                uint32_t st, en;
                derive_one(P, s_lut, raw[j], isz[j], mps[j], tile_pe, cur0.x, cur0.y, cur0.z < cur0.w, lut[j], st, en, qs[j], qe[j], q[j], uq[j]);
                odd = odd || (q[j] && ((int32_t)st != qs[j] || (int32_t)en != qe[j]));
            }
            anyq = anyq || q[j];
        }
        if (__ballot(odd)) {                                                    // rare
            anyq = false;
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                const int32_t t = raw[j].tid;
                uint4 tr = make_uint4(0xffffffffu, 0, 0, 0);
                uint32_t bb = 0;
                if (ex[j] && t >= 0 && t < P.n_tid) {
                    tr = *reinterpret_cast<const uint4 *>(&P.tidrec[t]);
                    bb = P.tidrec[t].bin_base;
                }
                if (FIRST) {
                    qs[j] = imax32(raw[j].pos, 0);
                    qe[j] = imin32(raw[j].tmpend, (int32_t)tr.y);
                    if (ex[j] && (int32_t)tr.x >= 0 && tr.z < tr.w && qs[j] < qe[j]) hit[j] = itx_first_lane(T, tr.z, bb, qs[j], qe[j]);
                } else {
                    uint32_t st, en;
                    derive_one(P, s_lut, raw[j], isz[j], mps[j], tile_pe, tr.x, tr.y, tr.z < tr.w, lut[j], st, en, qs[j], qe[j], q[j], uq[j]);
                    if (q[j]) classify_global(T, P, tr.z, bb, qs[j], qe[j], st, en, hit[j], sA[j], sB[j], hB[j]);
                }
                q[j] = false;
            }
        }
        // ---- cnt[0..7] (generic.c:1048-1055)
        {
            uint32_t n4 = 0;                                                            // 3-bit fields, each <= RPL
#pragma unroll
            for (int j = 0; j < RPL; j++) n4 += lut[j];
            accA += n4 & 0x1c71c7u;
            accB += (n4 >> 3) & 0x1c71c7u;
            if (++tiles == 15) {                                                        // 15 * 4 < 64
#pragma unroll
                for (int k = 0; k < 4; k++) acc[k] += ((accA >> (6 * k)) & 63u) | (((accB >> (6 * k)) & 63u) << 16);
                accA = accB = tiles = 0;
            }
        }
        ITX_ABLATE_AT(2, (uint32_t)qs[0] + (uint32_t)qe[1] + accA + accB)

        // ---- classify: every record left is on the cached reference
        if (__ballot(anyq)) {
            uint32_t lo_w = 0, wn = 0;
            uint2 bs = make_uint2(0, 0);
            int32_t mn = 0x7fffffff, mx = 0;
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                mn = q[j] ? imin32(qs[j], mn) : mn;
                mx = q[j] ? imax32(qe[j], mx) : mx;
            }
            mn = wave_min_i32(mn);
            mx = wave_max_i32(mx);
            const uint32_t bin_lo = (uint32_t)mn >> T.shift;
            const uint32_t nb = ((uint32_t)mx >> T.shift) + 1 - bin_lo + 1;         // bins bin_lo .. bin(mx)+1
            bool fast = nb <= 128;
            const bool wide = nb > 64;                                              // wave-uniform
            uint32_t bsx_hi = 0;
            if (fast) {
                if (lane < nb) bs = T.bl[cur_bb + bin_lo + lane];
                if (wide && lane + 64 < nb) bsx_hi = T.bl[cur_bb + bin_lo + 64 + lane].x;
                lo_w = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)bs.y);
                const uint32_t hi_w = wide ? (uint32_t)__builtin_amdgcn_readlane((int32_t)bsx_hi, (int)(nb - 65))
                                           : (uint32_t)__builtin_amdgcn_readlane((int32_t)bs.x, (int)(nb - 1));
                wn = hi_w > lo_w ? hi_w - lo_w : 0u;
                fast = wn <= ITX_WIN;
            }
            if (fast) {
                if (wn) {
                    const uint4 *src = reinterpret_cast<const uint4 *>(T.iv + lo_w);
                    for (uint32_t k = lane; k < 2 * wn; k += 64) win[2 + k] = src[k];   // coalesced, 16 B per lane
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    ITX_ABLATE_AT(3, win[2 * (lane % wn)].x + wn)
                    // Candidates of a record: window rows below the top of qe's bin (one lane read of the index
                    // slice); rows starting at or after qe fail the overlap test by themselves. The four records of a
                    // lane walk down in lockstep, entry kk then kk-1 ..., and the wave goes on while any record's current
                    // row says rows below it still end past the query start (pbelow > qs; binRange.c:209-225 without the
                    // bin lists). pbelow only shrinks on the way down and a row that ends at or before qs cannot overlap,
                    // so records that are done just keep stepping, count nothing, and come to rest on the sentinel.
                    uint32_t kk[RPL], hk[RPL];        // hk: window entries of the hits so far, one byte each, newest lowest
                    uint32_t f_rk[RPL];               // FIRST: smallest list-order rank among the hits so far (hk = its entry)
#ifdef ITX_SMALL_SLICE
                    if (nb <= 4) {
                        // dense read sets: the tile spans a handful of bins and the slice's entries are wave-uniform values —
                        // three lane reads (scalar registers) and three selects per record instead of a trip through the LDS
                        // crossbar (ds_bpermute) that the scan would have to wait for
                        const uint32_t e1 = __builtin_elementwise_sub_sat((uint32_t)__builtin_amdgcn_readlane((int32_t)bs.x, 1), lo_w);
                        const uint32_t e2 = __builtin_elementwise_sub_sat((uint32_t)__builtin_amdgcn_readlane((int32_t)bs.x, 2), lo_w);
                        const uint32_t e3 = __builtin_elementwise_sub_sat((uint32_t)__builtin_amdgcn_readlane((int32_t)bs.x, 3), lo_w);
#pragma unroll
                        for (int j = 0; j < RPL; j++) {
                            const uint32_t b = ((uint32_t)qe[j] >> T.shift) - bin_lo + 1u;         // 1 .. nb - 1 for a record that goes on
                            const uint32_t top = b == 1u ? e1 : b == 2u ? e2 : e3;
                            kk[j] = q[j] ? top : 0u;
                            hk[j] = 0;
                            f_rk[j] = 0xffffffffu;
                        }
                    } else
#endif
                    {
#pragma unroll
                    for (int j = 0; j < RPL; j++) {
                        const uint32_t top = top_entry(bs.x, bsx_hi, wide, qe[j], T.shift, bin_lo, lo_w);   // a shuffle: every lane takes part
                        kk[j] = q[j] ? top : 0u;                  // rows [0, top) <=> entries [1, top]
                        hk[j] = 0;
                        f_rk[j] = 0xffffffffu;
                    }
                    }
                    bool any;
                    do {
                        uint4 v[RPL];
#pragma unroll
                        for (int j = 0; j < RPL; j++) v[j] = win[2 * kk[j]];              // s, e, pbelow, rank
                        any = false;
#pragma unroll
                        for (int j = 0; j < RPL; j++) {
                            const int32_t ov = clip_ov((int32_t)v[j].x, (int32_t)v[j].y, qs[j], qe[j]);
                            if (FIRST) {
                                const bool better = ov > 0 && v[j].w < f_rk[j];
                                f_rk[j] = better ? v[j].w : f_rk[j];
                                hk[j] = better ? kk[j] : hk[j];
                            } else {
                                hk[j] = ov > 0 ? (hk[j] << 8) | kk[j] : hk[j];
                            }
                            any = any || (int32_t)v[j].z > qs[j];
                            kk[j] = __builtin_elementwise_sub_sat(kk[j], 1u);
                        }
                    } while (__ballot(any));
                    if (FIRST) {
#pragma unroll
                        for (int j = 0; j < RPL; j++) hit[j] = hk[j] ? (int32_t)hk[j] - 1 + (int32_t)lo_w : -1;
                        __builtin_amdgcn_wave_barrier();                                    // the window is rewritten next tile
                        goto classified;
                    }
                    // Best hit (generic.c:950-970): in binKeeperFind's list order, the LAST hit whose coverage exceeds the
                    // previous hit's. All hits of a record share the denominator (end - start) and, for overlaps below
                    // 2^23, distinct integer overlaps give distinct f32 quotients — so with two hits the pick is an integer
                    // comparison on (rank, overlap): the one that comes first in list order stays unless the other overlaps
                    // more. With one hit the other is the sentinel (negative overlap, last rank) and the same lines hold.
                    // The -c test is one multiply (itx_cov_bounds). Three or more hits, giant fragments and quotients within
                    // 2^-20 of -c replay the reference's arithmetic in full.
                    bool anyrare = false;
#pragma unroll
                    for (int j = 0; j < RPL; j++) {
                        const uint32_t h_k = hk[j] & 0xffu, g_k = (hk[j] >> 8) & 0xffu;
                        const uint4 vh = win[2 * h_k], vg = win[2 * g_k];
                        const int32_t ov_h = clip_ov((int32_t)vh.x, (int32_t)vh.y, qs[j], qe[j]);
                        const int32_t ov_g = clip_ov((int32_t)vg.x, (int32_t)vg.y, qs[j], qe[j]);
                        const bool take_h = ov_h > ov_g || (vh.w < vg.w && ov_h == ov_g);
                        const uint32_t ck = take_h ? h_k : g_k;
                        const uint32_t qlen = (uint32_t)qe[j] - (uint32_t)qs[j];
                        const bool simple = hk[j] != 0 && hk[j] < 0x10000u && qlen < (1u << 23);
                        const float ovf = (float)imax32(ov_h, ov_g), qf = (float)qlen;
                        const bool pass = ovf >= qf * P.cov_hi, fail = ovf < qf * P.cov_lo;
                        hit[j] = (simple && pass) ? (int32_t)ck : -1;
                        anyrare = anyrare || (hk[j] != 0 && !(simple && (pass || fail)));
                    }
                    if (__ballot(anyrare)) {
                        IvLds A{win + 2};
                        uint32_t tops[RPL];                       // from the index in memory: the slice registers are long dead
#pragma unroll
                        for (int j = 0; j < RPL; j++)
                            tops[j] = (hk[j] != 0 && (hk[j] >= 0x10000u || (uint32_t)qe[j] - (uint32_t)qs[j] >= (1u << 23)))
                                          ? __builtin_elementwise_sub_sat(T.bl[cur_bb + ((uint32_t)qe[j] >> T.shift) + 1].x, lo_w)
                                          : 0u;
#pragma unroll
                        for (int j = 0; j < RPL; j++) {
                            const uint32_t qlen = (uint32_t)qe[j] - (uint32_t)qs[j];
                            if (hk[j] != 0 && (hk[j] >= 0x10000u || qlen >= (1u << 23))) {
                                const uint32_t top = tops[j];
                                uint32_t low = top;                                        // rows [low, top) can overlap
                                while (low > 0) {
                                    const int32_t pb = (int32_t)win[2 * low].z;
                                    low--;
                                    if (pb <= qs[j]) break;
                                }
                                const int32_t r = itx_pick_multi(A, low, top, qs[j], qe[j], (uint32_t)qs[j], (uint32_t)qe[j], P.min_cov);
                                hit[j] = r >= 0 ? r + 1 : -1;
                            } else if (hk[j] != 0) {
                                const uint32_t h_k = hk[j] & 0xffu, g_k = (hk[j] >> 8) & 0xffu;
                                const uint4 vh = win[2 * h_k], vg = win[2 * g_k];
                                const int32_t ov_h = clip_ov((int32_t)vh.x, (int32_t)vh.y, qs[j], qe[j]);
                                const int32_t ov_g = clip_ov((int32_t)vg.x, (int32_t)vg.y, qs[j], qe[j]);
                                const bool take_h = ov_h > ov_g || (vh.w < vg.w && ov_h == ov_g);
                                const float c = __fdiv_rn((float)imax32(ov_h, ov_g), (float)qlen);     // generic.c:296-301
                                hit[j] = !(c < P.min_cov) ? (int32_t)(take_h ? h_k : g_k) : -1;       // generic.c:961-962
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < RPL; j++) {
                        const uint32_t k = hit[j] >= 0 ? (uint32_t)hit[j] : 0u;
                        const uint4 v0 = win[2 * k], v1 = win[2 * k + 1];
                        const ItxIv r = {(int32_t)v0.x, (int32_t)v0.y, (int32_t)v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                        uint32_t first;
                        const uint32_t nc = itx_cov_range(r, (uint32_t)qs[j], (uint32_t)qe[j], &first);
                        hB[j] = hit[j] >= 0 && nc != 0;
                        sA[j] = hB[j] ? first : r.zslot;
                        sB[j] = first + nc;
                        hit[j] = hit[j] >= 0 ? hit[j] - 1 + (int32_t)lo_w : -1;
                    }
                    __builtin_amdgcn_wave_barrier();                                    // the window is rewritten next tile
                }
            } else {                                                            // sparse or unsorted records: global lookups
#pragma unroll
                for (int j = 0; j < RPL; j++)
                    if (q[j]) {
                        if (FIRST) hit[j] = itx_first_lane(T, cur0.z, cur_bb, qs[j], qe[j]);
                        else classify_global(T, P, cur0.z, cur_bb, qs[j], qe[j], (uint32_t)qs[j], (uint32_t)qe[j], hit[j], sA[j], sB[j], hB[j]);
                    }
            }
        }
    classified:
        ITX_ABLATE_AT(4, (uint32_t)(hit[0] + hit[1]))

        // ---- hits (generic.c:1030-1032), per wave
        unsigned long long mA[RPL];
#pragma unroll
        for (int j = 0; j < RPL; j++) {
            mA[j] = __ballot(hit[j] >= 0);
            n_hit += (uint32_t)__popcll(mA[j]);
            n_hitu += (uint32_t)__popcll(mA[j] & __ballot(uq[j]));
        }
        ITX_ABLATE_AT(5, (uint32_t)(hit[0] + hit[1]) + n_hit)

        // ---- chosen rows back to the caller (row ids as passed to itx_table_create)
        if (d_hit_row) {
            int32_t h[RPL];
#pragma unroll
            for (int j = 0; j < RPL; j++) h[j] = hit[j] >= 0 ? T.orig[hit[j]] : -1;
            if (full) {
#if RPL == 4
                *reinterpret_cast<int4 *>(d_hit_row + r0) = make_int4(h[0], h[1], h[2], h[3]);
#else
                *reinterpret_cast<int2 *>(d_hit_row + r0) = make_int2(h[0], h[1]);
#endif
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
                    const uint32_t unit = T.row_unit[hit[j]];
                    atomicAdd((unsigned long long *)&u64[16 + unit], 1ull);
                    if (hB[j]) {
                        atomicAdd(&u32[L.d_all + sA[j]], 1u);
                        atomicAdd(&u32[L.d_all + sB[j]], 0xffffffffu);              // -1 mod 2^32
                    }
                    if (uq[j]) {
                        atomicAdd((unsigned long long *)&u64[16 + L.n_units + unit], 1ull);
                        if (hB[j]) {
                            atomicAdd(&u32[L.d_uniq + sA[j]], 1u);
                            atomicAdd(&u32[L.d_uniq + sB[j]], 0xffffffffu);
                        }
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
        } else if (EMIT) {
            // One 8-byte key per classified record: low word = type | uniq << 2 | len << 3, high word = slot.
            //   type 0: a start mark at slot and an end mark at slot + len (both inside one partition, len < 2^13: a longer
            //           range inside a wide partition goes as two keys as well)
            //   type 1: a start mark only (the read adds no coverage, or its end mark lies in another partition)
            //   type 2: an end mark only (the other half of such a read)
            // The keys leave in RECORD order (neighbouring records mostly hit the same row, so the partition path
            // downstream sees long runs of one partition): every wave fills its own quarter of the workgroup's region
            // front to back, a lane's keys right behind those of the lanes below it (DPP prefix sum) — 8-byte stores
            // that together cover one contiguous stretch per tile. No cursor is shared, nothing is waited for.
            bool two[RPL];
            uint32_t c = 0;
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                // (a coverage range of 2^13 slots or more inside one WIDE partition goes as two keys as well: the 4-byte keys of
                // k_scatter hold 13 bits of length)
                two[j] = WIDE ? hB[j] & ((((sA[j] ^ sB[j]) >> E.log_w) != 0u) | (sB[j] - sA[j] >= (1u << ITX_LOGW)))
                              : hB[j] && (sA[j] >> E.log_w) != (sB[j] >> E.log_w);
                c += (hit[j] >= 0 ? 1u : 0u) + (two[j] ? 1u : 0u);
            }
            uint32_t inc = c;                                                    // inclusive prefix sum over the lanes
            ITX_DPP_STEP(uadd32, inc, 0, 0x111, 0xf);
            ITX_DPP_STEP(uadd32, inc, 0, 0x112, 0xf);
            ITX_DPP_STEP(uadd32, inc, 0, 0x114, 0xf);
            ITX_DPP_STEP(uadd32, inc, 0, 0x118, 0xf);
            ITX_DPP_STEP(uadd32, inc, 0, 0x142, 0xa);
            ITX_DPP_STEP(uadd32, inc, 0, 0x143, 0xc);
            uint32_t at = w_keys + inc - c;
            w_keys += (uint32_t)__builtin_amdgcn_readlane((int32_t)inc, 63);     // wave-uniform
#ifdef ITX_LANE_RUNS
            // keys per partition of this workgroup's region: a lane's four consecutive records mostly chose the same row, so
            // their start marks fall into one partition — one LDS add per RUN of equal partitions inside the lane instead of
            // one per key (the adds of a wave go to a handful of addresses and serialise there)
            uint32_t pa[RPL];
#pragma unroll
            for (int j = 0; j < RPL; j++) pa[j] = sA[j] >> E.log_w;
            uint32_t run = 0;
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                const bool v = hit[j] >= 0;
                run += v ? 1u : 0u;
                const bool flush = v && (j == RPL - 1 || hit[j + 1 < RPL ? j + 1 : j] < 0 || pa[j + 1 < RPL ? j + 1 : j] != pa[j]);
                if (flush) atomicAdd(&s_pc[pa[j]], run);
                run = flush ? 0u : run;
            }
#endif
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                const uint32_t u = uq[j] ? 4u : 0u;
                const uint32_t lo = (hB[j] && !two[j]) ? (((sB[j] - sA[j]) << 3) | u) : (1u | u);
                if (hit[j] >= 0) {
                    if (ITX_ABLATE_NOT(7)) w_out[at] = make_uint2(lo, sA[j]);
#ifndef ITX_LANE_RUNS
                    if (ITX_ABLATE_NOT(6)) atomicAdd(&s_pc[sA[j] >> E.log_w], 1u);   // keys per partition of this workgroup's region
#endif
                }
                at += hit[j] >= 0 ? 1u : 0u;
                if (two[j]) {
                    if (ITX_ABLATE_NOT(7)) w_out[at] = make_uint2(2u | u, sB[j]);
                    if (ITX_ABLATE_NOT(6)) atomicAdd(&s_pc[sB[j] >> E.log_w], 1u);
                }
                at += two[j] ? 1u : 0u;
            }
        }
    }
#ifdef ITX_ABLATE
    if (sink == 0x7fffff01u) s_cnt[12] = sink;
#endif
    if (WHAT != ITX_DO_CLASSIFY && !FIRST) {              // classify-only launches leave every accumulator alone
        uint32_t tot[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            acc[k] += ((accA >> (6 * k)) & 63u) | (((accB >> (6 * k)) & 63u) << 16);
            tot[2 * k] = wave_sum_u32(acc[k] & 0xffffu);
            tot[2 * k + 1] = wave_sum_u32(acc[k] >> 16);
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (tot[k]) atomicAdd(&s_cnt[k], tot[k]);
            if (tot[7]) atomicAdd(&s_cnt[11], tot[7]);    // reads_nonredundant_unique == reads_mapped_unique without -R
            if (n_hit) atomicAdd(&s_cnt[9], n_hit);
            if (n_hitu) atomicAdd(&s_cnt[10], n_hitu);
        }
    }
    __syncthreads();
    if (WHAT != ITX_DO_CLASSIFY && !FIRST && threadIdx.x < 16 && s_cnt[threadIdx.x])
        atomicAdd((unsigned long long *)&u64[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    if (EMIT) {
        if (lane == 0) blk_cnt[4 * blockIdx.x + w] = w_keys;
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

unsigned itx_stream_blocks(int device, size_t n)
{
    if (const char *s = getenv("ITX_STREAM_BLOCKS")) {
        const long v = atol(s);
        if (v >= 1 && v <= 32768) return (unsigned)v;
    }
    static int cus_of[64];                                  // 0: not asked yet
    int cus = device >= 0 && device < 64 ? cus_of[device] : 0;
    if (cus <= 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
        if (device >= 0 && device < 64) cus_of[device] = cus;
    }
    unsigned one = (unsigned)cus * ITX_LB;                  // what the chip holds at once
    if (one > 2048u) one = 2048u;
    size_t target = ITX_STREAM_REGION;
    if (const char *s = getenv("ITX_STREAM_REGION")) {
        const long v = atol(s);
        if (v >= 1024) target = (size_t)v;
    }
    size_t max_rounds = ITX_STREAM_ROUNDS;
    if (const char *s = getenv("ITX_STREAM_ROUNDS")) {
        const long v = atol(s);
        if (v >= 1 && v <= 64) max_rounds = (size_t)v;
    }
    size_t rounds = (n + (size_t)one * target / 2) / ((size_t)one * target);
    rounds = rounds < 1 ? 1 : rounds > max_rounds ? max_rounds : rounds;
    return one * (unsigned)rounds;
}

void itx_stream_plan(unsigned blocks, size_t n, size_t *span, unsigned *n_blocks)
{
    size_t sp = (n + blocks - 1) / blocks;
    sp = (sp + ITX_STREAM_TILE - 1) / ITX_STREAM_TILE * ITX_STREAM_TILE;
    if (sp == 0) sp = ITX_STREAM_TILE;
    *span = sp;
    *n_blocks = (unsigned)((n + sp - 1) / sp);
}

int itx_launch_stream(int what, const ItxDevTable &T, const ItxRunParams &P, const ItxDevBatch &B, size_t n, size_t span,
                      unsigned n_blocks, int32_t *d_hit_row, uint64_t *u64, uint32_t *u32, const ItxAccumLayout &L, uint2 *keys0,
                      uint32_t *blk_cnt, const ItxEmitPlan &E, hipStream_t st)
{
    if (n == 0) return ITX_OK;
    const uintptr_t al = (uintptr_t)B.tid | (uintptr_t)B.pos | (uintptr_t)B.tmpend | (uintptr_t)d_hit_row | (uintptr_t)B.mpos | (uintptr_t)B.isize;
    if ((al & 15u) || (((uintptr_t)B.mapq | (uintptr_t)B.flag5) & 3u)) {
        itx_set_error("record arrays must be 16-byte aligned (tid/pos/tmpend/mpos/isize/hit_row) and 4-byte aligned (mapq/flag5)");
        return ITX_E_ARG;
    }
    if (span % ITX_STREAM_TILE || span > (size_t)16000 * ITX_STREAM_TILE) {         // 16-bit per-lane counters: <= 65535 / RPL tiles per wave
        itx_set_error("internal: span %zu is not a multiple of %u or too long", span, ITX_STREAM_TILE);
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
    case ITX_DO_FIND_FIRST:
        hipLaunchKernelGGL(k_stream<ITX_DO_FIND_FIRST>, g, b, 0, st, T, P, B, n, span, d_hit_row, u64, u32, L, keys0, blk_cnt, E);
        break;
    case ITX_DO_EMIT:
        if (E.log_w > ITX_LOGW)
            hipLaunchKernelGGL(k_stream<ITX_DO_EMIT_WIDE>, g, b, (size_t)E.n_part * 4, st, T, P, B, n, span, d_hit_row, u64, u32, L, keys0,
                               blk_cnt, E);
        else
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
